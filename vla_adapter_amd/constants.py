"""Mirror of prismatic/vla/constants.py (token constants :11-15, per-platform shapes :28-54, argv sniffing :58-75)."""
import sys

IGNORE_INDEX = -100
ACTION_TOKEN_BEGIN_IDX = 151386
STOP_INDEX = 2
NUM_TOKENS = 64

PLATFORM_CONSTANTS = {
    "LIBERO": dict(NUM_ACTIONS_CHUNK=8, ACTION_DIM=7, PROPRIO_DIM=8, ACTION_PROPRIO_NORMALIZATION_TYPE="bounds_q99"),
    "CALVIN": dict(NUM_ACTIONS_CHUNK=8, ACTION_DIM=7, PROPRIO_DIM=8, ACTION_PROPRIO_NORMALIZATION_TYPE="bounds_q99"),
    "ALOHA": dict(NUM_ACTIONS_CHUNK=25, ACTION_DIM=14, PROPRIO_DIM=14, ACTION_PROPRIO_NORMALIZATION_TYPE="bounds"),
    "BRIDGE": dict(NUM_ACTIONS_CHUNK=5, ACTION_DIM=7, PROPRIO_DIM=7, ACTION_PROPRIO_NORMALIZATION_TYPE="bounds_q99"),
}


def detect_robot_platform(argv=None) -> str:
    """Same substring sniffing of the command line as the reference (constants.py:58-75); default LIBERO."""
    cmd = " ".join(sys.argv if argv is None else argv).lower()
    for key in ("libero", "aloha", "bridge", "calvin"):
        if key in cmd:
            return key.upper()
    return "LIBERO"


ROBOT_PLATFORM = detect_robot_platform()
NUM_ACTIONS_CHUNK = PLATFORM_CONSTANTS[ROBOT_PLATFORM]["NUM_ACTIONS_CHUNK"]
ACTION_DIM = PLATFORM_CONSTANTS[ROBOT_PLATFORM]["ACTION_DIM"]
PROPRIO_DIM = PLATFORM_CONSTANTS[ROBOT_PLATFORM]["PROPRIO_DIM"]
ACTION_PROPRIO_NORMALIZATION_TYPE = PLATFORM_CONSTANTS[ROBOT_PLATFORM]["ACTION_PROPRIO_NORMALIZATION_TYPE"]
