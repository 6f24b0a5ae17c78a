// Library identity and device query.
#include "ipx_common.h"
#include <string.h>
#include <stdio.h>

static thread_local char g_last_error[256] = "";

void ipx_note_error(hipError_t e, const char *file, int line) {
  snprintf(g_last_error, sizeof(g_last_error), "%s (%s:%d)", hipGetErrorString(e), file, line);
}

extern "C" {

const char *ipx_version(void) { return "ipx 0.1 (gfx950)"; }

const char *ipx_last_error(void) { return g_last_error; }

int ipx_device_info(int *cu_count, int *lds_bytes, char *arch, int arch_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return IPX_ELAUNCH;
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) return IPX_ELAUNCH;
  if (cu_count) *cu_count = p.multiProcessorCount;
  if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
  if (arch && arch_len > 0) {
    strncpy(arch, p.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return IPX_OK;
}

}  // extern "C"
