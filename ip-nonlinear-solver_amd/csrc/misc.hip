// Library identity and device query.
#include "ipx_common.h"
#include <string.h>
#include <stdio.h>
#include <atomic>
#include <chrono>

constexpr int IPX_READ_MAX = 512;
extern "C" int ipx_read_doubles(const double *dev, int k, double *host_out, void *stream);

// ipx_read_doubles: k doubles into host-coherent memory as 16-byte granules that validate
// themselves -- (low word, tag, high word, tag), ONE write-through store each, the tag = the
// read's sequence number: no fence, no flag behind the data (a system-scope release at this
// point would first write back every dirty L2 line the loop before it left: +18 us per read
// behind a batch of CG iterations, measured with a fence + flag version)
typedef unsigned int ipx_u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(IPX_READ_MAX)
k_publish(const double *__restrict__ src, int k, ipx_u4 *dst, unsigned int tag) {
  if ((int)threadIdx.x >= k) return;
  const unsigned long long bits = (unsigned long long)__double_as_longlong(src[threadIdx.x]);
  ipx_u4 w;
  w.x = (unsigned)(bits & 0xffffffffull); w.y = tag;
  w.z = (unsigned)(bits >> 32);           w.w = tag;
  ipx_u4 *d = dst + threadIdx.x;
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d), "v"(w) : "memory");
}

static thread_local char g_last_error[256] = "";
static std::atomic<long long> g_ipx_reads{0};

void ipx_note_error(hipError_t e, const char *file, int line) {
  snprintf(g_last_error, sizeof(g_last_error), "%s (%s:%d)", hipGetErrorString(e), file, line);
}

// the calling thread's pinned granule buffer and the next tag
int ipx_read_begin(unsigned int **pinned_out, unsigned int *tag_out) {
  static thread_local unsigned int *pinned = nullptr;      // [IPX_READ_MAX] granules of 4 words
  static thread_local unsigned int seq = 0;
  if (!pinned) {
    if (hipHostMalloc((void **)&pinned, IPX_READ_MAX * 16, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
      pinned = nullptr;
      return IPX_ENOMEM;
    }
    memset(pinned, 0, IPX_READ_MAX * 16);
  }
  if (++seq == 0) ++seq;                                   // (0: the buffer's initial tags)
  *pinned_out = pinned;
  *tag_out = seq;
  return IPX_OK;
}

// poll the first k granules for `tag`, copy the values out
int ipx_read_wait(unsigned int *pinned, unsigned int seq, int k, double *host_out, hipStream_t st) {
  volatile unsigned int *w = pinned;
  g_ipx_reads.fetch_add(1, std::memory_order_relaxed);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  int done = 0;                                            // granules [0, done) have arrived
  while (true) {
    while (done < k && w[4 * done + 1] == seq && w[4 * done + 3] == seq) ++done;
    if (done == k) break;
    __builtin_ia32_pause();
    if ((++spins & 0xfffu) == 0 &&
        std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
      hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
      for (done = 0; done < k && w[4 * done + 1] == seq && w[4 * done + 3] == seq; ++done) {}
      if (done != k) return IPX_ELAUNCH;
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  for (int i = 0; i < k; ++i) {
    const unsigned long long bits = (unsigned long long)w[4 * i] | ((unsigned long long)w[4 * i + 2] << 32);
    memcpy(host_out + i, &bits, sizeof(double));
  }
  return IPX_OK;
}

// ipx_read_folded: scalar q = the fold of descs.d[q].count partials (ipx_sum_partials: the
// order of the reductions' own second stage, same bits), published like k_publish's
struct FoldDescs { ipx_fold_desc d[IPX_FOLD_MAX]; };
__global__ void __launch_bounds__(IPX_BLOCK)
k_publish_folded(FoldDescs descs, int nd, ipx_u4 *dst, unsigned int tag) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  for (int q = 0; q < nd; ++q) {
    const double *part = descs.d[q].part;
    const int count = descs.d[q].count, op = descs.d[q].op;
    double r;
    if (op == IPX_MAX) r = ipx_sum_partials<IPX_MAX>(part, count, lds);
    else if (op == IPX_MIN) r = ipx_sum_partials<IPX_MIN>(part, count, lds);
    else r = ipx_sum_partials<IPX_SUM>(part, count, lds);
    if (threadIdx.x == 0) {
      const unsigned long long bits = (unsigned long long)__double_as_longlong(r);
      ipx_u4 w;
      w.x = (unsigned)(bits & 0xffffffffull); w.y = tag;
      w.z = (unsigned)(bits >> 32);           w.w = tag;
      ipx_u4 *d = dst + q;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d), "v"(w) : "memory");
    }
    __syncthreads();                                  // (lds is reused by the next fold)
  }
}

// ipx_fold_combine: out[0] = ((w0 r0 + w1 r1) + w2 r2) + ... with r_q folded as above -- a
// scalar a later kernel consumes (the user's objective handed to ipx_sqp_judge) never visits
// the host
struct FoldWeights { double w[IPX_FOLD_MAX]; };
__global__ void __launch_bounds__(IPX_BLOCK)
k_fold_combine(FoldDescs descs, FoldWeights wts, int nd, double *__restrict__ out) {
  __shared__ double lds[IPX_BLOCK / IPX_WAVE];
  double acc = 0.0;
  for (int q = 0; q < nd; ++q) {
    const double *part = descs.d[q].part;
    const int count = descs.d[q].count, op = descs.d[q].op;
    double r;
    if (op == IPX_MAX) r = ipx_sum_partials<IPX_MAX>(part, count, lds);
    else if (op == IPX_MIN) r = ipx_sum_partials<IPX_MIN>(part, count, lds);
    else r = ipx_sum_partials<IPX_SUM>(part, count, lds);
    const double term = wts.w[q] * r;
    acc = q == 0 ? term : acc + term;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = acc;
}

long long g_ipx_launches = 0;

extern "C" {

const char *ipx_version(void) { return "ipx 0.1 (gfx950)"; }

long long ipx_launch_count(void) { return g_ipx_launches; }

long long ipx_read_count(void) { return g_ipx_reads.load(); }

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
// `stream`: a one-workgroup kernel stores them into a pinned, host-coherent buffer of the
// library's (one per host thread) as tagged granules (k_publish); the host polls the tags.  No
// runtime synchronisation call on the way (hipMemcpy into pageable memory: 21 us per
// read behind a small kernel, hipMemcpyAsync into pinned memory + hipStreamSynchronize: 18 --
// the outer loops' scalar reads are the host's largest single item of a solve).  A read that
// has not arrived after 2 s falls back to hipStreamSynchronize (and reports its error).
int ipx_read_doubles(const double *dev, int k, double *host_out, void *stream) {
  if (!dev || !host_out || k < 0 || k > IPX_READ_MAX) return IPX_EINVAL;
  if (k == 0) return IPX_OK;
  unsigned int *pinned; unsigned int tag;
  int rc = ipx_read_begin(&pinned, &tag);
  if (rc != IPX_OK) return rc;
  hipLaunchKernelGGL(k_publish, dim3(1), dim3(IPX_READ_MAX), 0, (hipStream_t)stream, dev, k,
                     (ipx_u4 *)pinned, tag);
  ++g_ipx_launches;
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
  }
  return ipx_read_wait(pinned, tag, k, host_out, (hipStream_t)stream);
}

// Blocking read-back of nd <= IPX_FOLD_MAX scalars, each the fold of a partial array a
// reduction's first stage left (ipx_dot_partials / ipx_norms_partials; op: IPX_SUM 0, IPX_MAX 1,
// IPX_MIN 2 -- the numbering of csrc/ipx_common.h): the fold runs in the read's own publish
// kernel, in the order of ipx_dot / ipx_norms' second launch (same bits) -- one launch less per
// reduction whose result only the host wants.
int ipx_read_folded(int nd, const ipx_fold_desc *descs, double *host_out, void *stream) {
  if (!descs || !host_out || nd < 0 || nd > IPX_FOLD_MAX) return IPX_EINVAL;
  if (nd == 0) return IPX_OK;
  FoldDescs D;
  for (int q = 0; q < nd; ++q) {
    if (!descs[q].part || descs[q].count < 1) return IPX_EINVAL;
    D.d[q] = descs[q];
  }
  for (int q = nd; q < IPX_FOLD_MAX; ++q) D.d[q] = descs[0];
  unsigned int *pinned; unsigned int tag;
  int rc = ipx_read_begin(&pinned, &tag);
  if (rc != IPX_OK) return rc;
  hipLaunchKernelGGL(k_publish_folded, dim3(1), dim3(IPX_BLOCK), 0, (hipStream_t)stream, D, nd,
                     (ipx_u4 *)pinned, tag);
  ++g_ipx_launches;
  {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
  }
  return ipx_read_wait(pinned, tag, nd, host_out, (hipStream_t)stream);
}

// The weighted sum of nd <= IPX_FOLD_MAX folded scalars into device memory (no read): out[0] =
// ((w[0] r_0 + w[1] r_1) + w[2] r_2) + ..., r_q folded as ipx_read_folded folds it -- what the
// host would compute from the values it read, bit for bit, for expressions written that way.
int ipx_fold_combine(int nd, const ipx_fold_desc *descs, const double *weights, double *out,
                     void *stream) {
  if (!descs || !weights || !out || nd < 1 || nd > IPX_FOLD_MAX) return IPX_EINVAL;
  FoldDescs D;
  FoldWeights W;
  for (int q = 0; q < IPX_FOLD_MAX; ++q) {
    const int k = q < nd ? q : 0;
    if (!descs[k].part || descs[k].count < 1) return IPX_EINVAL;
    D.d[q] = descs[k];
    W.w[q] = weights[k];
  }
  hipLaunchKernelGGL(k_fold_combine, dim3(1), dim3(IPX_BLOCK), 0, (hipStream_t)stream, D, W, nd,
                     out);
  ++g_ipx_launches;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
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

