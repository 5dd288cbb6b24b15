// Error plumbing + version of the C ABI.  No exceptions cross the boundary: every entry returns an int and
// records a thread-local message retrievable with vla_last_error().
#include "common.h"
#include "../../include/vla_native.h"
#include <string.h>

static thread_local char g_err[256] = "";

extern "C" void vla_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* vla_last_error(void) { return g_err; }
extern "C" int vla_version(void) { return VLA_ABI_VERSION; }
extern "C" int vla_desc_size(int which) {
  switch (which) {
    case 0: return (int)sizeof(vla_gemm_desc);
    case 1: return (int)sizeof(vla_attn_desc);
    case 2: return (int)sizeof(vla_head_attn_desc);
    case 3: return (int)sizeof(vla_gemm_tn_desc);
    default: return -1;
  }
}
