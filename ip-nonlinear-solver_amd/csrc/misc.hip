// Library identity and device query.
#include "ipx_common.h"
#include <string.h>
#include <stdio.h>
#include <atomic>
#include <chrono>

constexpr int IPX_READ_MAX = 512;
extern "C" int ipx_read_doubles(const double *dev, int k, double *host_out, void *stream);

// ipx_read_doubles: k doubles into host-coherent memory, the sequence word behind them
__global__ void __launch_bounds__(IPX_READ_MAX)
k_publish(const double *__restrict__ src, int k, double *dst, unsigned long long *word,
          unsigned long long seq) {
  if ((int)threadIdx.x < k) dst[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

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

// Blocking read-back of k <= IPX_READ_MAX doubles of device memory behind everything queued on
// `stream`: a one-workgroup kernel copies them into a pinned, host-coherent buffer of the
// library's (one per host thread) and stores a sequence number behind them; the host polls that
// word.  No runtime synchronisation call on the way (hipMemcpy into pageable memory: 21 us per
// read behind a small kernel, hipMemcpyAsync into pinned memory + hipStreamSynchronize: 18 --
// the outer loops' scalar reads are the host's largest single item of a solve).  A read that
// has not arrived after 2 s falls back to hipStreamSynchronize (and reports its error).
int ipx_read_doubles(const double *dev, int k, double *host_out, void *stream) {
  static thread_local double *pinned = nullptr;      // [IPX_READ_MAX] values, then the sequence word
  static thread_local unsigned long long seq = 0;
  if (!dev || !host_out || k < 0 || k > IPX_READ_MAX) return IPX_EINVAL;
  if (k == 0) return IPX_OK;
  if (!pinned) {
    if (hipHostMalloc((void **)&pinned, (IPX_READ_MAX + 1) * sizeof(double),
                      hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      pinned = nullptr;
      return IPX_ENOMEM;
    }
    memset(pinned, 0, (IPX_READ_MAX + 1) * sizeof(double));
  }
  hipStream_t st = (hipStream_t)stream;
  volatile unsigned long long *word = (volatile unsigned long long *)(pinned + IPX_READ_MAX);
  ++seq;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(IPX_READ_MAX), 0, st, dev, k, pinned,
                     (unsigned long long *)(pinned + IPX_READ_MAX), seq);
  if (hipGetLastError() != hipSuccess) return IPX_ELAUNCH;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (*word != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xfffu) == 0 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
      if (*word != seq) return IPX_ELAUNCH;
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  memcpy(host_out, pinned, (size_t)k * sizeof(double));
  return IPX_OK;
}

}  // extern "C"

// Inside the library: k ints behind the same mechanism (the pivot flags of a factorization).
int ipx_read_ints(const int *dev, int k, int *host_out, hipStream_t st) {
  if (k < 0 || k > 2 * IPX_READ_MAX) return IPX_EINVAL;
  double tmp[IPX_READ_MAX];
  // (an odd count reads one int past the end: the callers' buffers are allocations of whole
  // doubles -- asserted where they are made)
  int rc = ipx_read_doubles((const double *)dev, (k + 1) / 2, tmp, st);
  if (rc == IPX_OK) memcpy(host_out, tmp, (size_t)k * sizeof(int));
  return rc;
}

