// Peer mailboxes for the row-sharded projected-CG loop (ipsolver/sharded.py): the ranks of
// one node exchange their few scalars per iteration (p'Hp; ||x+ap||^2, #violations, ||g||^2,
// ||A g||^2: the reductions of qp_subproblem.py:556,583,626) and the halo of g by writing
// straight into each other's HBM over xGMI -- no collective call and no host between the
// iterations of a batch.
//
// Set-up (host, once per group): every rank allocates its mailbox (uncached device memory),
// exports a hipIpc handle, the handles travel through the launcher's side channel
// (torch.distributed.all_gather_object in ipsolver/sharded.py), every rank maps the others'.
// In the loop the mailboxes are touched by kernels only (csrc/cg.hip k_cg_pack_comm, LL
// words: ipx_common.h).  Works the same between processes that share one device (the
// rehearsal topology of the tests) and between the GPUs of a node.
#include "ipx_common.h"
#include <string.h>
#include <new>

namespace {

constexpr long long IPX_PEER_HOST_PATIENCE = 16;

// all-reduce(sum) of nq doubles and nothing else: the latency probe of bench.py
__global__ void __launch_bounds__(64)
k_peer_allreduce(ipx_peer_view pv, uint32_t seq, int nq, const double *__restrict__ in,
                 double *__restrict__ out, int *__restrict__ failed) {
  __shared__ double vals[IPX_MAX_PEERS * IPX_PEER_NQ];
  const int tid = threadIdx.x, slot = seq & (IPX_PEER_SLOTS - 1);
  const long long deadline = (long long)wall_clock64() + pv.timeout_ticks;
  for (int i = tid; i < pv.world * nq; i += blockDim.x) {
    const int r = i / nq, q = i - r * nq;
    ipx_ll_store(pv.mbox[r] + ipx_peer_scal_word(slot, pv.rank, q), in[q], seq);
  }
  bool ok = true;
  for (int i = tid; i < pv.world * nq; i += blockDim.x) {
    const int r = i / nq, q = i - r * nq;
    double v = 0.0;
    ok = ipx_ll_load(pv.mbox[pv.rank] + ipx_peer_scal_word(slot, r, q), seq, v, deadline) && ok;
    vals[r * IPX_PEER_NQ + q] = v;
  }
  if (!ok) *failed = 1;
  __syncthreads();
  if (tid < nq) {
    double s = 0.0;
    for (int r = 0; r < pv.world; ++r) s += vals[r * IPX_PEER_NQ + tid];   // rank order: same bits everywhere
    out[tid] = s;
  }
}

// all-gather of nq <= 7 doubles per rank, the rank's own passed BY VALUE (no upload on the way):
// out[r * nq + q] = rank r's q-th value, then two failure words.  The
// outer loops' scalar collectives (a pack of norms and dot products: summed / maximised over
// the ranks by the host, in rank order -- the same bits on every rank) without a call into
// torch.distributed.
struct ipx_vals8 { double v[IPX_PEER_NQ]; };
__global__ void __launch_bounds__(128)
k_peer_allgather(ipx_peer_view pv, uint32_t seq, int nq, ipx_vals8 mine, double *__restrict__ out,
                 int *__restrict__ failed) {
  // (word nq of every rank: its sticky failure word -- a halo exchange of ITS that timed out
  // -- so that the whole group learns of it at the same collective)
  const int tid = threadIdx.x, slot = seq & (IPX_PEER_SLOTS - 1), nw = nq + 1;
  // Sixteen times the patience of a wait inside the loop's kernels: the host's collectives run
  // BETWEEN those kernels, and a rank whose kernels are just sitting out their own deadlines
  // (up to eight times the base one: the resident form's commit hop) is late, not gone.  A wait
  // that gave up while the late rank then found this rank's words would split the group -- one
  // rank on the mailboxes, one on the fall-back transport, each waiting for the other.
  const long long deadline = (long long)wall_clock64() + IPX_PEER_HOST_PATIENCE * pv.timeout_ticks;
  __shared__ int bad, before;
  const double failed_mine = (double)*failed;
  if (tid == 0) { bad = 0; before = 0; }
  __syncthreads();
  for (int i = tid; i < pv.world * nw; i += blockDim.x) {
    const int r = i / nw, q = i - r * nw;
    double v = mine.v[0];
#pragma unroll
    for (int k = 1; k < IPX_PEER_NQ; ++k) v = q == k ? mine.v[k] : v;
    if (q == nq) v = failed_mine;
    ipx_ll_store(pv.mbox[r] + ipx_peer_scal_word(slot, pv.rank, q), v, seq);
  }
  bool ok = true;
  for (int i = tid; i < pv.world * nw; i += blockDim.x) {
    const int r = i / nw, q = i - r * nw;
    double v = 0.0;
    ok = ipx_ll_load(pv.mbox[pv.rank] + ipx_peer_scal_word(slot, r, q), seq, v, deadline) && ok;
    if (q < nq) out[r * nq + q] = v;
    else if (v != 0.0) before = 1;
  }
  if (!ok) bad = 1;
  __syncthreads();
  if (tid == 0) {
    // [world nq]: an EARLIER halo exchange timed out on some rank -- what it left in the halos
    // cannot be repaired: every rank raises; [world nq + 1]: this collective's own wait did --
    // the group repeats it on another transport
    out[pv.world * nq] = (double)before;
    out[pv.world * nq + 1] = (double)bad;
  }
}

// halo exchange of up to four segments of a local vector v (each [seg_lo | own_lo .. own_hi |
// seg_hi)): this rank's first / last own entries go into the neighbours' halo areas, its own
// halo entries are taken from what the neighbours stored here -- the conventions of the loop's
// own kernels (csrc/cg.hip k_cg_step2_hp PEER: areas by parity of the halo sequence number).
__global__ void __launch_bounds__(IPX_BLOCK)
k_peer_exchange(ipx_peer_job pj, double *__restrict__ v, int *__restrict__ failed) {
  const ipx_peer_view &pv = pj.pv;
  const int tid = threadIdx.x, par = pj.hseq & 1;
  if (pv.rank > 0)
    for (int k = 0; k < pj.nseg; ++k)
      for (int j = tid; j < pj.send_left[k]; j += IPX_BLOCK)
        ipx_ll_store(pv.mbox[pv.rank - 1] + ipx_peer_halo_word(pv.cap, 1, par, pj.push_l[k] + j),
                     v[pj.own_lo[k] + j], pj.hseq);
  if (pv.rank < pv.world - 1)
    for (int k = 0; k < pj.nseg; ++k)
      for (int j = tid; j < pj.send_right[k]; j += IPX_BLOCK)
        ipx_ll_store(pv.mbox[pv.rank + 1] + ipx_peer_halo_word(pv.cap, 0, par, pj.push_r[k] + j),
                     v[pj.own_hi[k] - pj.send_right[k] + j], pj.hseq);
  // (the host collectives' patience, see k_peer_allgather: a neighbour that is merely late --
  // it sat out a wait of its own -- must not cost the halos, which nothing repairs)
  const long long deadline = (long long)wall_clock64() + IPX_PEER_HOST_PATIENCE * pv.timeout_ticks;
  const unsigned long long *mine = pv.mbox[pv.rank];
  bool ok = true;
  for (int k = 0; k < pj.nseg; ++k) {
    if (pv.rank > 0)
      for (int col = pj.seg_lo[k] + tid; col < pj.own_lo[k]; col += IPX_BLOCK) {
        double t = 0.0;
        ok = ipx_ll_load(mine + ipx_peer_halo_word(pv.cap, 0, par, pj.off_l[k] + (col - pj.seg_lo[k])),
                         pj.hseq, t, deadline) && ok;
        v[col] = t;
      }
    if (pv.rank < pv.world - 1)
      for (int col = pj.own_hi[k] + tid; col < pj.seg_hi[k]; col += IPX_BLOCK) {
        double t = 0.0;
        ok = ipx_ll_load(mine + ipx_peer_halo_word(pv.cap, 1, par, pj.off_r[k] + (col - pj.own_hi[k])),
                         pj.hseq, t, deadline) && ok;
        v[col] = t;
      }
  }
  if (!ok) *failed = 1;
}

// `reps` round trips of one tagged word between this rank and `partner` (the lower rank
// serves): store into the partner's mailbox, spin on the own one.  ticks[0] = wall-clock ticks
// (100 MHz) of the whole exchange as this rank saw it, ticks[1] = 1 when a wait timed out.
// What one cross-GPU hand-off of the resident loop kernel costs, measured on the spot.
__global__ void __launch_bounds__(64)
k_peer_pingpong(ipx_peer_view pv, int partner, uint32_t seq0, int reps, long long *ticks) {
  if (threadIdx.x != 0) return;
  const bool serve = pv.rank < partner;
  const long long t0 = (long long)wall_clock64();
  bool ok = true;
  for (int i = 0; i < reps && ok; ++i) {
    uint32_t seq = seq0 + (uint32_t)i;
    if (seq == 0) seq = 1;                          // (never the tag of a zeroed word)
    const int slot = seq & (IPX_PEER_SLOTS - 1);
    unsigned long long *out = pv.mbox[partner] + ipx_peer_scal_word(slot, pv.rank, IPX_PEER_NQ - 1);
    const unsigned long long *in = pv.mbox[pv.rank] + ipx_peer_scal_word(slot, partner, IPX_PEER_NQ - 1);
    const long long deadline = (long long)wall_clock64() + pv.timeout_ticks;
    double v = 0.0;
    if (serve) {
      ipx_ll_store(out, (double)i, seq);
      ok = ipx_ll_load(in, seq, v, deadline);
    } else {
      ok = ipx_ll_load(in, seq, v, deadline);
      ipx_ll_store(out, v, seq);
    }
  }
  ticks[0] = (long long)wall_clock64() - t0;
  ticks[1] = ok ? 0 : 1;
}

}  // namespace

extern "C" {

// `reps` round trips with `partner` (-1: this rank sits the round out).  COLLECTIVE in its
// bookkeeping: every rank of the group calls it once per round -- the sequence numbers advance
// by `reps` on every rank, playing or not.  ticks2: device array of two int64.
int ipx_peer_pingpong(void *peer, int32_t partner, int32_t reps, long long *ticks2, void *stream) {
  if (!peer || reps < 1 || reps > (1 << 20)) return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  if (partner >= p->view.world || partner == p->view.rank) return IPX_EINVAL;
  const uint32_t seq0 = p->seq + 1;
  p->seq += (uint32_t)reps;
  if (p->seq == 0) p->seq = 1;
  if (partner < 0) return IPX_OK;
  if (!ticks2 || !ipx_peer_ready(peer)) return IPX_EINVAL;
  hipLaunchKernelGGL(k_peer_pingpong, dim3(1), dim3(64), 0, (hipStream_t)stream, p->view,
                     (int)partner, seq0, (int)reps, ticks2);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

int ipx_peer_handle_bytes(void) { return (int)sizeof(hipIpcMemHandle_t); }

void *ipx_peer_create(int32_t rank, int32_t world, int64_t halo_cap) {
  if (world < 1 || world > IPX_MAX_PEERS || rank < 0 || rank >= world || halo_cap < 0) return nullptr;
  ipx_peer *p = new (std::nothrow) ipx_peer();
  if (!p) return nullptr;
  memset(p, 0, sizeof(*p));
  p->view.rank = rank;
  p->view.world = world;
  p->view.cap = halo_cap;
  p->view.timeout_ticks = IPX_PEER_TIMEOUT_TICKS;
  p->seq = p->hseq = 0;
  p->bytes = 8 * ((int64_t)IPX_PEER_SCAL_WORDS + 2 * 2 * halo_cap * 2);
  void *mem = nullptr;
  // uncached: remote writes must be seen by the spinning reader, local polls must not be
  // served from a stale L2 line
  hipError_t e = hipExtMallocWithFlags(&mem, (size_t)p->bytes, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&mem, (size_t)p->bytes, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) {
    ipx_note_error(e, __FILE__, __LINE__);
    delete p;
    return nullptr;
  }
  if (hipMemset(mem, 0, (size_t)p->bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(mem);
    delete p;
    return nullptr;
  }
  p->view.mbox[rank] = (unsigned long long *)mem;
  return p;
}

int ipx_peer_export(void *peer, void *handle_out) {
  if (!peer || !handle_out) return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p->view.mbox[p->view.rank]);
  if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
  memcpy(handle_out, &h, sizeof(h));
  return IPX_OK;
}

int ipx_peer_import(void *peer, int32_t rank, const void *handle_in) {
  if (!peer || !handle_in) return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  if (rank < 0 || rank >= p->view.world || rank == p->view.rank || p->view.mbox[rank]) return IPX_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle_in, sizeof(h));
  void *mem = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&mem, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
  p->opened[rank] = mem;
  p->view.mbox[rank] = (unsigned long long *)mem;
  return IPX_OK;
}

// 1 when every rank's mailbox is mapped
int ipx_peer_ready(void *peer) {
  if (!peer) return 0;
  ipx_peer *p = (ipx_peer *)peer;
  for (int r = 0; r < p->view.world; ++r)
    if (!p->view.mbox[r]) return 0;
  return 1;
}

// How long a kernel waits for a peer's word before it raises stop code 7 (default 10 s: a peer
// that is merely slow -- a time-shared GPU, a lazily loaded code object, a profiler -- is not a
// dead one).
int ipx_peer_set_timeout(void *peer, double seconds) {
  if (!peer || !(seconds > 0.0) || seconds > 3600.0) return IPX_EINVAL;
  ((ipx_peer *)peer)->view.timeout_ticks = (long long)(seconds * 1e8);
  return IPX_OK;
}

int64_t ipx_peer_halo_capacity(void *peer) { return peer ? ((ipx_peer *)peer)->view.cap : 0; }

// scalar / halo exchanges issued so far (identical on every rank of a healthy group)
int ipx_peer_sequence(void *peer, int64_t *out2) {
  if (!peer || !out2) return IPX_EINVAL;
  out2[0] = ((ipx_peer *)peer)->seq;
  out2[1] = ((ipx_peer *)peer)->hseq;
  return IPX_OK;
}

// loop kernels launched so far that did their part of a collective themselves (the PEER forms
// of csrc/cg.hip: ipx_shard2_ext.fuse_comm)
int64_t ipx_peer_fused_launches(void *peer) { return peer ? ((ipx_peer *)peer)->fused : 0; }

// ---- hand-off buffers of the resident loop kernel's PEER form (csrc/resident.hip)
int ipx_peer_attach_resident(void *peer, int64_t words) {
  if (!peer || words < 1) return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  if (p->res[p->view.rank]) return IPX_EINVAL;
  void *mem = nullptr;
  hipError_t e = hipExtMallocWithFlags(&mem, (size_t)words * 8, hipDeviceMallocUncached);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    e = hipExtMallocWithFlags(&mem, (size_t)words * 8, hipDeviceMallocFinegrained);
  }
  if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ENOMEM; }
  if (hipMemset(mem, 0, (size_t)words * 8) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipFree(mem);
    return IPX_ELAUNCH;
  }
  p->res[p->view.rank] = (unsigned long long *)mem;
  p->res_words = words;
  p->rseq = 1;
  return IPX_OK;
}

int ipx_peer_export_resident(void *peer, void *handle_out) {
  if (!peer || !handle_out) return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  if (!p->res[p->view.rank]) return IPX_EINVAL;
  hipIpcMemHandle_t h;
  hipError_t e = hipIpcGetMemHandle(&h, p->res[p->view.rank]);
  if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
  memcpy(handle_out, &h, sizeof(h));
  return IPX_OK;
}

int ipx_peer_import_resident(void *peer, int32_t rank, const void *handle_in) {
  if (!peer || !handle_in) return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  if (rank < 0 || rank >= p->view.world || rank == p->view.rank || p->res[rank]) return IPX_EINVAL;
  hipIpcMemHandle_t h;
  memcpy(&h, handle_in, sizeof(h));
  void *mem = nullptr;
  hipError_t e = hipIpcOpenMemHandle(&mem, h, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) { ipx_note_error(e, __FILE__, __LINE__); return IPX_ELAUNCH; }
  p->res_opened[rank] = mem;
  p->res[rank] = (unsigned long long *)mem;
  return IPX_OK;
}

// builds the device-side table of the buffers; 1 when every rank's is mapped
int ipx_peer_resident_ready(void *peer) {
  if (!peer) return 0;
  ipx_peer *p = (ipx_peer *)peer;
  for (int r = 0; r < p->view.world; ++r)
    if (!p->res[r]) return 0;
  if (!p->res_tab) {
    void *tab = nullptr;
    if (hipMalloc(&tab, sizeof(unsigned long long *) * IPX_MAX_PEERS) != hipSuccess) return 0;
    if (hipMemcpy(tab, p->res, sizeof(unsigned long long *) * IPX_MAX_PEERS, hipMemcpyHostToDevice) !=
        hipSuccess) {
      (void)hipFree(tab);
      return 0;
    }
    p->res_tab = (unsigned long long **)tab;
  }
  return 1;
}

int64_t ipx_peer_resident_launches(void *peer) { return peer ? ((ipx_peer *)peer)->res_launches : 0; }

void ipx_peer_destroy(void *peer) {
  if (!peer) return;
  ipx_peer *p = (ipx_peer *)peer;
  for (int r = 0; r < p->view.world; ++r)
    if (p->res_opened[r]) (void)hipIpcCloseMemHandle(p->res_opened[r]);
  if (p->res[p->view.rank]) (void)hipFree(p->res[p->view.rank]);
  if (p->res_tab) (void)hipFree(p->res_tab);
  for (int r = 0; r < p->view.world; ++r)
    if (p->opened[r]) (void)hipIpcCloseMemHandle(p->opened[r]);
  if (p->view.mbox[p->view.rank]) (void)hipFree(p->view.mbox[p->view.rank]);
  delete p;
}

// All-gather of nq <= 7 host scalars per rank (vals: nq doubles of HOST memory, passed to the
// kernel by value): out (device, world * nq + 1 doubles) receives every rank's values in rank
// order and, last, two failure words: an earlier halo exchange timed out (failed: device int,
// sticky, set by ipx_peer_exchange) / this all-gather did.  Collective: every rank calls it with the same nq.
int ipx_peer_allgather(void *peer, int32_t nq, const double *vals, double *out, int *failed,
                       void *stream) {
  if (!peer || nq < 1 || nq > IPX_PEER_NQ - 1 || !vals || !out || !failed || !ipx_peer_ready(peer))
    return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  ipx_vals8 mine{};
  for (int q = 0; q < nq; ++q) mine.v[q] = vals[q];
  if (++p->seq == 0) ++p->seq;
  hipLaunchKernelGGL(k_peer_allgather, dim3(1), dim3(128), 0, (hipStream_t)stream, p->view, p->seq,
                     (int)nq, mine, out, failed);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// Halo exchange of nseg <= 4 segments of the local vector v.  geom: 6 ints per segment --
// seg_lo, own_lo, own_hi, seg_hi (local indices), send_left, send_right (own entries the left /
// right neighbour keeps as halo).  Collective; never synchronises; a wait that times out sets
// *failed (device int, sticky).
int ipx_peer_exchange(void *peer, double *v, int32_t nseg, const int64_t *geom, int *failed,
                      void *stream) {
  if (!peer || !v || nseg < 1 || nseg > 4 || !geom || !failed || !ipx_peer_ready(peer))
    return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  ipx_peer_job pj;
  memset(&pj, 0, sizeof(pj));
  pj.pv = p->view;
  pj.nseg = nseg;
  int64_t inl = 0, inr = 0, outl = 0, outr = 0;
  for (int k = 0; k < nseg; ++k) {
    const int64_t *g = geom + 6 * k;
    pj.seg_lo[k] = (int)g[0]; pj.own_lo[k] = (int)g[1]; pj.own_hi[k] = (int)g[2];
    pj.seg_hi[k] = (int)g[3]; pj.send_left[k] = (int)g[4]; pj.send_right[k] = (int)g[5];
    if (g[0] > g[1] || g[1] > g[2] || g[2] > g[3] || g[4] < 0 || g[5] < 0 || g[4] > g[2] - g[1] ||
        g[5] > g[2] - g[1])
      return IPX_EINVAL;
    pj.off_l[k] = (int)inl; pj.off_r[k] = (int)inr; pj.push_l[k] = (int)outl; pj.push_r[k] = (int)outr;
    inl += g[1] - g[0]; inr += g[3] - g[2]; outl += g[4]; outr += g[5];
  }
  if (inl > p->view.cap || inr > p->view.cap || outl > p->view.cap || outr > p->view.cap)
    return IPX_EINVAL;
  if (++p->hseq == 0) ++p->hseq;
  pj.hseq = p->hseq;
  pj.seq = p->seq;
  hipLaunchKernelGGL(k_peer_exchange, dim3(1), dim3(IPX_BLOCK), 0, (hipStream_t)stream, pj, v,
                     failed);
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

// `reps` all-reduces of nq <= 8 doubles, back to back on `stream` (in / out: device arrays;
// failed: device int, set when a wait timed out).  The measured floor of the mailbox path.
int ipx_peer_allreduce(void *peer, int32_t nq, const double *in, double *out, int *failed,
                       int32_t reps, void *stream) {
  if (!peer || nq < 1 || nq > IPX_PEER_NQ || !in || !out || !failed || !ipx_peer_ready(peer))
    return IPX_EINVAL;
  ipx_peer *p = (ipx_peer *)peer;
  for (int i = 0; i < reps; ++i) {
    if (++p->seq == 0) ++p->seq;
    hipLaunchKernelGGL(k_peer_allreduce, dim3(1), dim3(64), 0, (hipStream_t)stream, p->view, p->seq,
                       (int)nq, in, out, failed);
  }
  IPX_CHECK_LAUNCH();
  return IPX_OK;
}

}  // extern "C"
