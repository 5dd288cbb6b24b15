"""MI355X-native VLA-Adapter fine-tune hot path (hand-written HIP kernels behind a C ABI + PyTorch host shims)."""
__version__ = "0.1.0"
