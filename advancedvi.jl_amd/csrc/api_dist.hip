// libmivi C ABI, part 3: sharded estimates -- RCCL behind the ABI, the peer-to-peer exchange areas, the
// dependent-chain and the pipelined batch of estimates over the ranks of one node.
#include "api_common.h"

// ---------------------------------------------------------------------------------------------
// Sharded finalisation and the collective behind the C ABI: reduce-scatter -> slice finalise -> all-gather ->
// unpack.  RCCL is opened with dlopen: a host without it (or a CPU-only symbol check) still loads libmivi.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
void load_rccl(RcclApi &api) {
  const char *env = getenv("MIVI_RCCL_LIB");
  const char *cands[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int pass = 0; pass < 2 && !api.lib; ++pass)       // pass 0: a copy the process already loaded (e.g. the host framework's)
    for (const char *n : cands)
      if (n && !api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
  if (!api.lib) return;
  api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
  api.ReduceScatter = (decltype(api.ReduceScatter))dlsym(api.lib, "ncclReduceScatter");
  api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
  api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
  api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.ReduceScatter || !api.AllGather) api.lib = nullptr;
}
// the fully built table behind a C++11 magic static: contexts initialised from different host threads see it complete or not at all
struct RcclOnce { RcclApi api; RcclOnce() { load_rccl(api); } };
RcclApi *rccl() {
  static RcclOnce once;
  return once.api.lib ? &once.api : nullptr;
}
long long slice_len_of(const mivi_ctx *c, int world) {
  const long long L = mivi_partials_len(c);
  return (L + world - 1) / world;
}
}  // namespace

int64_t mivi_slice_len(const mivi_ctx_t *c, int32_t world) { return (c && world > 0) ? slice_len_of(c, world) : 0; }

mivi_status_t mivi_finalize_slice(mivi_ctx_t *c, const void *params, const void *slice_sum, int32_t rank, int32_t world, void *final_slice) {
  if (!c || !params || !slice_sum || !final_slice || world <= 0 || rank < 0 || rank >= world) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const long long n = slice_len_of(c, world);
  if (world > 1 && n < world + 2) return fail(c, MIVI_ERR_UNSUPPORTED, "parameter vector too short to shard over this many ranks");
  launch_finalize_slice(c, params, slice_sum, (long long)rank * n, n, final_slice);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_unpack_final(mivi_ctx_t *c, const void *packed_final, void *value, void *grad) {
  if (!c || !packed_final || !value || !grad) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  launch_unpack_final(c, packed_final, value, grad);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

mivi_status_t mivi_comm_unique_id(void *id_host) {
  if (!id_host) return MIVI_ERR_BAD_ARG;
  RcclApi *r = rccl();
  if (!r) return MIVI_ERR_UNSUPPORTED;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return MIVI_ERR_HIP;
  memcpy(id_host, &id, sizeof(id));
  return MIVI_OK;
}

// ---- peer-to-peer exchange buffers (kernels_p2p.hip) ---------------------------------------------------------------------------------
namespace {
struct P2PHandle {   // what travels between the ranks (MIVI_P2P_HANDLE_BYTES = 256 per rank)
  uint32_t magic, version;
  int32_t rank, world;
  int64_t L, n, cn;
  int32_t esize, G;
  uint64_t bytes, pid, local_ptr;
  int32_t device, pad;
  hipIpcMemHandle_t ipc;
};
static_assert(sizeof(P2PHandle) <= MIVI_P2P_HANDLE_BYTES, "handle blob");
constexpr uint32_t kP2PMagic = 0x4D495650u;   // "MIVP"
constexpr int kLanes = 1, kRing = 8, kGroup = 4;   // (kernels_p2p.hip: kP2PLanes, kP2PRing, kP2PGroup)
struct P2PTableHost { char *stage[kLanes][8]; char *fin[kLanes][8]; unsigned *arr[kLanes][8]; unsigned *farr[kLanes][8]; };   // == P2PTable (kernels_p2p.hip)

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// geometry of the exchange for (L, world): identical on every rank
void p2p_geometry(long long L, int R, long long &n, long long &cn, int &G, int &vs) {
  n = ((L + R - 1) / R + 3) & ~3LL;
  while ((L - 1) % n == 0) n += 4;        // the two scalars (L - 2, L - 1) must lie in ONE slice
  vs = (int)((L - 2) / n);
  // Small chunks, but at most 255 (+ the value workgroup = one workgroup per CU): the exchange kernel is persistent and spins beside the
  // compute chain.  More would be faster for an exchange on its own (system-scope accesses are limited per CU) but 513 spinning
  // workgroups held every CU's registers and the compute kernels could not be scheduled beside them at all (found on the GPU: the
  // hand-over timed out); measured in the pipelined batch (groups of four estimates, one lane): 127 -> 18.9, 191 -> 16.4, 255 -> 16.3,
  // 383 -> 17.6 us per estimate.
  long long g = (n + 511) / 512;
  G = (int)(g < 1 ? 1 : (g > 255 ? 255 : g));
  cn = ((n + G - 1) / G + 3) & ~3LL;
}
}  // namespace

// host-only: the geometry the exchange uses for a partial vector of length L over `world` ranks: out = {slice length n, chunk length cn,
// chunk workgroups G, value-owner rank}.  Identical on every rank by construction (tests/test_abi_and_host.py checks its invariants).
void mivi_p2p_geometry(int64_t L, int32_t world, int64_t *out4) {
  long long n, cn;
  int G, vs;
  p2p_geometry(L, world, n, cn, G, vs);
  out4[0] = n; out4[1] = cn; out4[2] = G; out4[3] = vs;
}

mivi_status_t mivi_p2p_detach(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->comm_stream2) (void)hipStreamSynchronize(c->comm_stream2);
  invalidate_graph(c);
  for (int r = 0; r < 8; ++r) {
    if (c->p2p_opened[r] && c->p2p_peer[r]) (void)hipIpcCloseMemHandle(c->p2p_peer[r]);
    c->p2p_opened[r] = false;
    c->p2p_peer[r] = nullptr;
  }
  if (c->p2p_buf) (void)hipFree(c->p2p_buf);
  c->p2p_buf = nullptr;
  c->p2p_bytes = 0;
  c->p2p_on = false;
  return MIVI_OK;
}

mivi_status_t mivi_p2p_export(mivi_ctx_t *c, int32_t rank, int32_t world, void *handle_out) {
  if (!c || !handle_out || world < 1 || world > 8 || rank < 0 || rank >= world) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  (void)mivi_p2p_detach(c);
  const long long L = mivi_partials_len(c);
  long long n, cn;
  int G, vs;
  p2p_geometry(L, world, n, cn, G, vs);
  const size_t es = c->esize;
  // per lane (double-buffered by epoch parity; an epoch carries a group of kGroup estimates): staging [2][V][R][n] T, final [2][V][R n] T,
  // arrival flags [2][R][G], final flags [2][R][G + 1]
  const size_t b_stage = align256((size_t)2 * kGroup * world * n * es), b_fin = align256((size_t)2 * kGroup * world * n * es);
  const size_t b_arr = align256((size_t)2 * world * G * 4), b_farr = align256((size_t)2 * world * (G + 1) * 4);
  const size_t lane_bytes = b_stage + b_fin + b_arr + b_farr;
  const size_t bytes = lane_bytes * kLanes;
  // FINE-GRAINED device memory: peers write it over xGMI, system-scope releases / acquires and the consumers' system-scope loads
  // (kernels_p2p.hip ld_sys) keep it coherent.  NOT hipDeviceMallocUncached: on this stack (ROCm 7.0 / gfx950) running the exchange on an
  // uncached allocation corrupted UNRELATED buffers of later contexts once the area had been freed and its pages re-used (found on one
  // GPU: the estimates of contexts created after a p2p context were off by 1e-2 until their buffers had been rewritten a few times;
  // fine-grained and plain allocations never showed it).
  void *buf = nullptr;
  hipError_t e = hipExtMallocWithFlags(&buf, bytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) { (void)hipGetLastError(); return fail(c, MIVI_ERR_HIP, "peer-to-peer exchange buffer: fine-grained allocation failed"); }
  if ((e = hipMemset(buf, 0, bytes)) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) {
    (void)hipFree(buf);   // (not yet owned by the context: nothing else would free it)
    c->err = std::string("peer-to-peer exchange buffer: ") + hipGetErrorString(e);
    return MIVI_ERR_HIP;
  }
  c->p2p_buf = buf;
  c->p2p_bytes = bytes;
  c->p2p_rank = rank; c->p2p_world = world; c->p2p_n = n; c->p2p_cn = cn; c->p2p_G = G; c->p2p_vs = vs;
  c->p2p_lane_bytes = lane_bytes; c->p2p_off_fin = b_stage; c->p2p_off_arr = b_stage + b_fin; c->p2p_off_farr = b_stage + b_fin + b_arr;
  P2PHandle h{};
  h.magic = kP2PMagic; h.version = 3; h.rank = rank; h.world = world; h.L = L; h.n = n; h.cn = cn; h.esize = (int32_t)es; h.G = G;
  h.bytes = bytes; h.pid = (uint64_t)getpid(); h.local_ptr = (uint64_t)(uintptr_t)buf; h.device = c->cfg.device;
  if (hipIpcGetMemHandle(&h.ipc, buf) != hipSuccess) {   // single-process use (tests, world = 1) still works through local_ptr
    (void)hipGetLastError();
    memset(&h.ipc, 0, sizeof(h.ipc));
    h.pad = 1;   // no IPC handle: other processes cannot attach
  }
  memset(handle_out, 0, MIVI_P2P_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  return MIVI_OK;
}

mivi_status_t mivi_p2p_attach(mivi_ctx_t *c, const void *handles) {
  if (!c || !handles) return MIVI_ERR_BAD_ARG;
  if (!c->p2p_buf) return fail(c, MIVI_ERR_BAD_ARG, "mivi_p2p_export has not been called");
  (void)hipSetDevice(c->cfg.device);
  const int R = c->p2p_world;
  const uint64_t me = (uint64_t)getpid();
  c->p2p_distinct = false;
  c->p2p_verified = false;
  P2PTableHost tab{};
  for (int r = 0; r < R; ++r) {
    P2PHandle h;
    memcpy(&h, (const char *)handles + (size_t)r * MIVI_P2P_HANDLE_BYTES, sizeof(h));
    if (h.magic != kP2PMagic || h.version != 3 || h.rank != r || h.world != R || h.L != mivi_partials_len(c) || h.n != c->p2p_n ||
        h.cn != c->p2p_cn || h.G != c->p2p_G || h.esize != (int32_t)c->esize || h.bytes != c->p2p_bytes)
      return fail(c, MIVI_ERR_BAD_ARG, "peer-to-peer handle does not match this context (rank order, family, d, dtype or world differ)");
    void *base = nullptr;
    if (r == c->p2p_rank) {
      base = c->p2p_buf;
    } else if (h.pid == me) {   // another context of this process: its pointer is valid here (peer access for another device)
      base = (void *)(uintptr_t)h.local_ptr;
      if (h.device != c->cfg.device) {
        const hipError_t e = hipDeviceEnablePeerAccess(h.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); return fail(c, MIVI_ERR_HIP, "hipDeviceEnablePeerAccess failed"); }
        (void)hipGetLastError();
      }
    } else {
      if (h.pad) return fail(c, MIVI_ERR_HIP, "peer exported no IPC handle (hipIpcGetMemHandle failed there)");
      if (hipIpcOpenMemHandle(&base, h.ipc, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        return fail(c, MIVI_ERR_HIP, "hipIpcOpenMemHandle failed (is HSA_ENABLE_IPC_MODE_LEGACY=0 exported?)");
      }
      c->p2p_opened[r] = true;
    }
    if (r != c->p2p_rank && (h.pid != me || h.device != c->cfg.device)) c->p2p_distinct = true;
    c->p2p_peer[r] = base;
    for (int ln = 0; ln < kLanes; ++ln) {
      char *lb = (char *)base + (size_t)ln * c->p2p_lane_bytes;
      tab.stage[ln][r] = lb;
      tab.fin[ln][r] = lb + c->p2p_off_fin;
      tab.arr[ln][r] = (unsigned *)(lb + c->p2p_off_arr);
      tab.farr[ln][r] = (unsigned *)(lb + c->p2p_off_farr);
    }
  }
  mivi_status_t s;
  if ((s = ensure(c, c->p2p_tab, sizeof(tab), false)) || (s = ensure(c, c->p2p_ctr, 512, false)) ||
      (s = ensure(c, c->p2p_scratch, ((size_t)mivi_params_len(c) + 4) * kGroup * kLanes * c->esize, false)))
    return s;
  HIPCHK(c, hipMemcpy(c->p2p_tab.p, &tab, sizeof(tab), hipMemcpyHostToDevice));
  {   // what the partial kernels need to store straight into the owners' staging areas (OutArgs::p2p_direct)
    P2PDirectTab dt{};
    for (int r = 0; r < R; ++r) dt.stage[r] = tab.stage[0][r];
    dt.n = c->p2p_n; dt.R = R; dt.rank = c->p2p_rank; dt.GV = kGroup;
    if ((s = ensure(c, c->p2p_direct, sizeof(dt), false))) return s;
    dt.ctr = (const unsigned *)c->p2p_ctr.p;
    HIPCHK(c, hipMemcpy(c->p2p_direct.p, &dt, sizeof(dt), hipMemcpyHostToDevice));
  }
  // (stream-ordered on the context's stream and waited for: a null-stream memset is NOT ordered against a non-blocking stream and
  //  would zero the epoch counter after the first exchange has advanced it)
  HIPCHK(c, hipMemsetAsync(c->p2p_ctr.p, 0, 512, c->stream));   // [lane] {epoch, ticket} at 64-byte spacing, ready at byte 256, freed[ring] at byte 320
  HIPCHK(c, hipStreamSynchronize(c->stream));
  invalidate_graph(c);
  c->p2p_on = true;
  return MIVI_OK;
}

mivi_status_t mivi_p2p_debug_words(mivi_ctx_t *c, uint32_t *out128) {   // developer: the exchange's device words (lane epochs, ready, freed)
  if (!c || !out128 || !c->p2p_ctr.p) return MIVI_ERR_BAD_ARG;
  HIPCHK(c, hipMemcpy(out128, c->p2p_ctr.p, 512, hipMemcpyDeviceToHost));
  return MIVI_OK;
}

// Diagnostics of the peer-to-peer exchange since the last reset (workgroup 0 of the exchange kernel; 100 MHz wall clock): out[0..2] = microseconds
// spent waiting for {the compute chain's hand-over, the peers' pushes (arrival flags), the owners' reduced chunks (final flags)}, out[3] = groups
// of estimates served, out[4] = payload bytes this rank stores into EACH peer per estimate (one slice pushed + one finished slice gathered),
// out[5] = slice length in elements.  A first multi-GPU run is diagnosable from these (bench.py --gpus N prints them per rank).
mivi_status_t mivi_p2p_stats(mivi_ctx_t *c, double *out6, int32_t reset) {
  if (!c || !out6) return MIVI_ERR_BAD_ARG;
  if (!c->p2p_on || !c->p2p_ctr.p) return fail(c, MIVI_ERR_BAD_ARG, "no peer-to-peer exchange buffers attached");
  (void)hipSetDevice(c->cfg.device);
  unsigned long long h[4] = {0, 0, 0, 0};
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(h, (unsigned *)c->p2p_ctr.p + 96, sizeof h, hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; ++k) out6[k] = (double)h[k] * 0.01;
  out6[3] = (double)h[3];
  out6[4] = 2.0 * (double)c->p2p_n * (double)c->esize;
  out6[5] = (double)c->p2p_n;
  if (reset) HIPCHK(c, hipMemset((unsigned *)c->p2p_ctr.p + 96, 0, sizeof h));
  return MIVI_OK;
}

mivi_status_t mivi_p2p_set_pipeline(mivi_ctx_t *c, int32_t on) {
  if (!c) return MIVI_ERR_BAD_ARG;
  // (a second persistent exchange kernel serving every other group was an option until the groups: two of them are 512 resident
  //  workgroups, which starve the compute chain of registers, and one measured better wherever both ran)
  if (on < 0 || on > 1) return fail(c, MIVI_ERR_BAD_ARG, "mivi_p2p_set_pipeline: 0 = off (serial steps), 1 = the persistent exchange kernel beside the compute chain");
  c->p2p_pipe_state = on ? 1 : -1;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_p2p_set_spin_budget(mivi_ctx_t *c, int32_t polls) {
  if (!c || polls < 16) return MIVI_ERR_BAD_ARG;
  c->p2p_spin = polls;
  invalidate_graph(c);
  return MIVI_OK;
}

mivi_status_t mivi_comm_set_route(mivi_ctx_t *c, int32_t route) {
  if (!c || route < 0 || route > 3) return MIVI_ERR_BAD_ARG;
  if (route == 3 && !c->p2p_on) return fail(c, MIVI_ERR_UNSUPPORTED, "peer-to-peer route: no exchange buffers attached (mivi_p2p_export / mivi_p2p_attach)");
  c->dist_route = route;
  invalidate_graph(c);
  return MIVI_OK;
}

int32_t mivi_comm_route(const mivi_ctx_t *c) {   // the route the next sharded estimate takes: 1 all-reduce, 2 reduce-scatter/all-gather, 3 peer-to-peer, 0 none (one rank, no communicator)
  if (!c) return 0;
  int r = c->dist_route;
  // automatic: across ranks the peer-to-peer kernel only once mivi_p2p_selfcheck has passed on every rank (an attach alone proves nothing
  // about the links); until then RCCL's reduce-scatter -> slice finalisation -> all-gather.  One rank (tests, world-1 runs): p2p as attached.
  if (r == 0) {
    const int world = c->p2p_on ? c->p2p_world : c->comm_world;
    if (c->p2p_on && (world == 1 || c->p2p_verified || !c->comm)) r = 3;   // (no communicator: the attached areas are the only exchange there is)
    else r = world > 1 ? 2 : (((size_t)mivi_partials_len(c) * c->esize >= ((size_t)16 << 20)) ? 2 : 1);
  }
  if (r == 3 && !c->p2p_on) r = 1;
  if ((r == 1 || r == 2) && !c->comm) return c->comm_world > 1 ? r : 0;
  return r;
}

mivi_status_t mivi_comm_destroy(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  (void)mivi_p2p_detach(c);
  if (c->comm) {
    RcclApi *r = rccl();
    (void)hipStreamSynchronize(c->stream);
    if (r) (void)r->CommDestroy((ncclComm_t)c->comm);
    c->comm = nullptr;
  }
  c->comm_rank = 0;
  c->comm_world = 1;
  return MIVI_OK;
}

mivi_status_t mivi_comm_init(mivi_ctx_t *c, const void *id_host, int32_t rank, int32_t world) {
  if (!c || world <= 0 || rank < 0 || rank >= world || (world > 1 && !id_host)) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  (void)mivi_comm_destroy(c);
  invalidate_graph(c);
  if (id_host) {   // also for world == 1: exercises the collective path on one GPU
    RcclApi *r = rccl();
    if (!r) return fail(c, MIVI_ERR_UNSUPPORTED, "librccl could not be opened (set MIVI_RCCL_LIB)");
    ncclUniqueId id;
    memcpy(&id, id_host, sizeof(id));
    ncclComm_t comm = nullptr;
    const ncclResult_t e = r->CommInitRank(&comm, world, id, rank);
    if (e != ncclSuccess) {
      c->err = std::string("ncclCommInitRank: ") + (r->GetErrorString ? r->GetErrorString(e) : "error");
      return MIVI_ERR_HIP;
    }
    c->comm = comm;
  }
  c->comm_rank = rank;
  c->comm_world = world;
  return MIVI_OK;
}

// Exchange the peer-to-peer handles through the RCCL communicator itself (hosts without another channel: julia/MIVI.jl) and attach.
// A failure leaves the context on the RCCL routes (the reason is in mivi_last_error).
mivi_status_t mivi_comm_enable_p2p(mivi_ctx_t *c) {
  if (!c) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  const int R = c->comm_world;
  if (R > 8) return fail(c, MIVI_ERR_UNSUPPORTED, "peer-to-peer exchange: at most 8 ranks (one xGMI node)");
  if (R > 1 && !c->comm) return fail(c, MIVI_ERR_BAD_ARG, "mivi_comm_init has not been called");
  std::vector<char> all((size_t)R * MIVI_P2P_HANDLE_BYTES);
  mivi_status_t s = mivi_p2p_export(c, c->comm_rank, R, all.data() + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES);
  if (s) return s;
  if (R > 1) {
    RcclApi *r = rccl();
    DevBuf tmp;
    if ((s = ensure(c, tmp, all.size(), false))) { (void)mivi_p2p_detach(c); return s; }
    auto give_up = [&](const char *why) {   // (the temporary and the exported areas go with every failure)
      (void)hipGetLastError();
      (void)hipFree(tmp.p);
      (void)mivi_p2p_detach(c);
      return fail(c, MIVI_ERR_HIP, why);
    };
    if (hipMemcpy((char *)tmp.p + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES, all.data() + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES,
                  MIVI_P2P_HANDLE_BYTES, hipMemcpyHostToDevice) != hipSuccess)
      return give_up("peer-to-peer handles: upload failed");
    const ncclResult_t e = r->AllGather((char *)tmp.p + (size_t)c->comm_rank * MIVI_P2P_HANDLE_BYTES, tmp.p, MIVI_P2P_HANDLE_BYTES, ncclChar,
                                        (ncclComm_t)c->comm, c->stream);
    if (e != ncclSuccess) return give_up("ncclAllGather of the peer-to-peer handles failed");
    if (hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(all.data(), tmp.p, all.size(), hipMemcpyDeviceToHost) != hipSuccess)
      return give_up("peer-to-peer handles: the gathered blobs could not be read back");
    (void)hipFree(tmp.p);
  }
  s = mivi_p2p_attach(c, all.data());
  if (s) { const std::string why = c->err; (void)mivi_p2p_detach(c); c->err = why; }
  return s;
}

// The peer-to-peer exchange against the RCCL all-reduce route on ONE sharded estimate, every rank collectively: both must give the same value and
// gradient (1e-5 relative; they differ by the summation order only), and every rank must say so (ncclAllReduce(min) of the verdicts) before the
// automatic route takes the peer-to-peer kernel.  rel_out (nullable): {value relative difference, gradient relative l2 difference, 1 if verified}.
// A peer-to-peer exchange that does not complete (bounded waits) detaches the areas: the context stays on the RCCL routes.
mivi_status_t mivi_p2p_selfcheck(mivi_ctx_t *c, const void *params, uint64_t idx, double *rel_out) {
  if (!c || !params) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  if (!c->p2p_on) return fail(c, MIVI_ERR_UNSUPPORTED, "mivi_p2p_selfcheck: no exchange areas attached");
  if (c->p2p_world > 1 && !c->comm) return fail(c, MIVI_ERR_BAD_ARG, "mivi_p2p_selfcheck: needs the RCCL communicator of mivi_comm_init to compare against");
  const size_t plen = (size_t)mivi_params_len(c), es = c->esize;
  DevBuf a, b;
  mivi_status_t s;
  if ((s = ensure(c, a, (plen + 16) * es, false)) || (s = ensure(c, b, (plen + 16) * es + 64, false))) { if (a.p) (void)hipFree(a.p); return s; }
  const int keep = c->dist_route;
  auto done = [&](mivi_status_t st) { (void)hipFree(a.p); (void)hipFree(b.p); c->dist_route = keep; invalidate_graph(c); return st; };
  c->dist_route = c->comm ? 1 : 3;
  invalidate_graph(c);
  // Every rank takes part in BOTH estimates and in the verdict's all-reduce WHATEVER happened to its own calls (bounded waits end the peer-to-peer
  // exchange on every rank): a rank that returned early -- also after a failed FIRST estimate -- would leave the others waiting inside RCCL.
  s = mivi_estimate_gradient_dist(c, params, idx, a.p, (char *)a.p + 16 * es);
  if (!s) s = mivi_synchronize(c);
  const mivi_status_t s_first = s;
  const std::string why_first = c->err;
  c->dist_route = 3;
  invalidate_graph(c);
  s = mivi_estimate_gradient_dist(c, params, idx, b.p, (char *)b.p + 16 * es);
  if (!s) s = mivi_synchronize(c);
  const mivi_status_t s_p2p = s_first ? s_first : s;
  const std::string why_p2p = s_first ? why_first : c->err;
  double rv = 1.0, rg = 1.0;
  int ok = 0;
  if (!s_p2p) {
    std::vector<char> ha((plen + 16) * es), hb((plen + 16) * es);
    if (hipMemcpy(ha.data(), a.p, ha.size(), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(hb.data(), b.p, hb.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      auto at = [&](const std::vector<char> &h, size_t i) { return es == 4 ? (double)((const float *)h.data())[i] : ((const double *)h.data())[i]; };
      const double v1 = at(ha, 0), v2 = at(hb, 0);
      double num = 0.0, den = 0.0;
      for (size_t i = 0; i < plen; ++i) { const double x = at(ha, 16 + i), y = at(hb, 16 + i); num += (y - x) * (y - x); den += x * x; }
      rv = fabs(v2 - v1) / (fabs(v1) > 0 ? fabs(v1) : 1.0);
      rg = sqrt(num) / (den > 0 ? sqrt(den) : 1.0);
      ok = (rv <= 1e-5 && rg <= 1e-5) ? 1 : 0;   // (NaN compares false)
    } else {
      (void)hipGetLastError();
    }
  }
  if (c->comm && c->comm_world > 1) {
    RcclApi *r = rccl();
    int *flag = (int *)((char *)b.p + (plen + 16) * es);
    if (!r || !r->AllReduce || hipMemcpy(flag, &ok, sizeof(int), hipMemcpyHostToDevice) != hipSuccess || r->AllReduce(flag, flag, 1, ncclInt32, ncclMin, (ncclComm_t)c->comm, c->stream) != ncclSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess || hipMemcpy(&ok, flag, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) {
      (void)mivi_p2p_detach(c);
      return done(fail(c, MIVI_ERR_HIP, "mivi_p2p_selfcheck: the ranks' verdicts could not be combined"));
    }
  }
  if (s_p2p) { (void)mivi_p2p_detach(c); c->err = why_p2p; return done(s_p2p); }
  if (!ok && c->p2p_world > 1) (void)mivi_p2p_detach(c);   // (some rank disagreed or failed: nobody keeps the areas, every rank stays on the RCCL routes)
  c->p2p_verified = ok == 1 && (c->p2p_distinct || c->p2p_world == 1);
  if (rel_out) { rel_out[0] = rv; rel_out[1] = rg; rel_out[2] = c->p2p_verified ? 1.0 : 0.0; }
  return done(MIVI_OK);
}

// buffers of the sharded estimate: padded partial vectors (two: the pipelined batch double-buffers them), slice sum, packed final
static mivi_status_t ensure_dist(mivi_ctx *c) {
  const int R = c->comm_world > c->p2p_world ? c->comm_world : c->p2p_world;
  const long long n = slice_len_of(c, c->comm_world), Lp = n * c->comm_world;
  long long need = Lp;
  if (c->p2p_on && c->p2p_n * c->p2p_world > need) need = c->p2p_n * c->p2p_world;
  (void)R;
  const size_t es = c->esize;
  const size_t need34 = c->p2p_on ? (size_t)need * es : 0;
  bool ring_short = false;
  for (int k = 0; k < 6; ++k) ring_short = ring_short || c->dist_ring[k].bytes < need34;
  if (c->dist_P.bytes < (size_t)need * es || c->dist_P2.bytes < (size_t)need * es || ring_short ||
      c->dist_S.bytes < (size_t)n * es || c->dist_F.bytes < (size_t)Lp * es) {
    invalidate_graph(c);
    mivi_status_t s;
    c->dist_P.bytes = 0; c->dist_P2.bytes = 0;   // (re-zero: the padding behind the partial vector must be 0)
    for (int k = 0; k < 6; ++k) {
      c->dist_ring[k].bytes = 0;
      if (need34 && (s = ensure(c, c->dist_ring[k], need34, true))) return s;
    }
    if ((s = ensure(c, c->dist_P, (size_t)need * es, true)) || (s = ensure(c, c->dist_P2, (size_t)need * es, true)) ||
        (s = ensure(c, c->dist_S, (size_t)n * es, true)) || (s = ensure(c, c->dist_F, (size_t)Lp * es, true)))
      return s;
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return MIVI_OK;
}

// The exchange + finalisation of ONE estimate whose partial vector sits in P, on c->stream: value / gradient on every rank.
static mivi_status_t dist_collective(mivi_ctx *c, const void *params, void *P, void *value, void *grad) {
  const int R = c->comm_world, rank = c->comm_rank;
  const size_t es = c->esize;
  const int route = mivi_comm_route(c);
  if (route == 3) {
    const void *Ps[1] = {P};
    launch_p2p_exchange(c, params, Ps, 1, value, grad, 7, 0, 1, 1, nullptr, nullptr, c->dist_direct);
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  const long long n = slice_len_of(c, R);
  // Route: two collectives cost one more launch + rendezvous than one; below 16 MB of partials the step is latency
  // bound and ONE all-reduce + the whole finalisation on every rank is the faster form (mivi_comm_set_route pins it).
  bool rsag = route == 2 || !c->comm;   // (one rank without a communicator: the slice kernels without the collectives)
  if (c->comm && !rsag && !rccl()->AllReduce) rsag = true;   // (a librccl without ncclAllReduce: the two-collective route needs only the required symbols)
  // the slice route gives every rank n >= world + 2 elements (the two trailing scalars must lie in the last slice); short parameter
  // vectors are exactly the ones the single all-reduce serves, so fall back to it instead of refusing
  if (rsag && R > 1 && n < R + 2) {
    if (c->comm && rccl()->AllReduce) rsag = false;
    else return fail(c, MIVI_ERR_UNSUPPORTED, "parameter vector too short to shard over this many ranks");
  }
  if (c->comm && !rsag) {
    RcclApi *r = rccl();
    const ncclDataType_t dt = c->cfg.dtype == MIVI_F32 ? ncclFloat : ncclDouble;
    if (r->AllReduce(P, P, (size_t)mivi_partials_len(c), dt, ncclSum, (ncclComm_t)c->comm, c->stream) != ncclSuccess)
      return fail(c, MIVI_ERR_HIP, "ncclAllReduce failed");
    launch_finalize(c, params, P, value, grad);
    HIPCHK(c, hipGetLastError());
    return MIVI_OK;
  }
  const void *sum = (const char *)P + (size_t)rank * n * es;   // one rank: its "slice" is the whole vector
  if (c->comm) {
    RcclApi *r = rccl();
    const ncclDataType_t dt = c->cfg.dtype == MIVI_F32 ? ncclFloat : ncclDouble;
    if (r->ReduceScatter(P, c->dist_S.p, (size_t)n, dt, ncclSum, (ncclComm_t)c->comm, c->stream) != ncclSuccess)
      return fail(c, MIVI_ERR_HIP, "ncclReduceScatter failed");
    sum = c->dist_S.p;
  }
  char *fin_slice = (char *)c->dist_F.p + (size_t)rank * n * es;
  launch_finalize_slice(c, params, sum, (long long)rank * n, n, fin_slice);
  if (c->comm) {
    RcclApi *r = rccl();
    const ncclDataType_t dt = c->cfg.dtype == MIVI_F32 ? ncclFloat : ncclDouble;
    if (r->AllGather(fin_slice, c->dist_F.p, (size_t)n, dt, (ncclComm_t)c->comm, c->stream) != ncclSuccess)
      return fail(c, MIVI_ERR_HIP, "ncclAllGather failed");
  }
  launch_unpack_final(c, c->dist_F.p, value, grad);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

// Direct staging (kernels_p2p.hip; DESIGN.md 7, cut (a)): on the peer-to-peer route the second-generation full-rank f32 kernels store every
// entry of the partial vector straight into its owner's staging area -- no ring slot, no push pass.  OFF by default (MIVI_P2P_DIRECT=1
// enables it): measured on one GPU (tools/dist_profile.py, north-star shape) the push pass it removes is worth 1.4 us of the exchange, but the
// packed triangle's column segments are not 16-byte aligned, so the VJP epilogue's stores become 4-byte system-scope stores (partial kernels
// 13.4 -> 17.2 us), and a staging slot is released only after the unpack (the compute chain runs one group ahead instead of two): serial step
// 36.9 -> 38.5 us, pipelined batch 13.2 -> 20.0 us per estimate.  It pays once the packed layout pads every column to 16 bytes.
static bool p2p_direct_ok(const mivi_ctx *c, const void *params) {
  static const bool off = !(getenv("MIVI_P2P_DIRECT") && atoi(getenv("MIVI_P2P_DIRECT")) == 1);
  if (off || !c->p2p_on || !c->p2p_direct.p || c->cfg.family != MIVI_FULLRANK || c->cfg.dtype != MIVI_F32 || c->dbg) return false;
  OutArgs on{};
  on.partials_mode = 1;
  return lds_route(c, params, c->cfg.n_mc, 1, on);
}

static mivi_status_t dist_check(mivi_ctx *c) {
  if (c->comm_world > 1 && !c->comm && !c->p2p_on) return fail(c, MIVI_ERR_BAD_ARG, "mivi_comm_init has not been called");
  if (c->p2p_on && (c->p2p_world != c->comm_world || c->p2p_rank != c->comm_rank) && (c->comm || c->comm_world > 1))
    return fail(c, MIVI_ERR_BAD_ARG, "peer-to-peer buffers were exported for another rank / world than the communicator's");
  return MIVI_OK;
}

// estimate_gradient! of ONE estimate whose n_mc * world samples are sharded over the ranks (this context draws columns
// [m_offset, m_offset + n_mc) of m_total): partials -> exchange -> finalisation, all on the context's stream.
mivi_status_t mivi_estimate_gradient_dist(mivi_ctx_t *c, const void *params, uint64_t idx, void *value, void *grad) {
  if (!c || !params || !value || !grad) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  if (c->p2p_on && !c->comm) { c->comm_world = c->p2p_world; c->comm_rank = c->p2p_rank; }   // (peer-to-peer without RCCL)
  mivi_status_t s;
  if ((s = dist_check(c)) || (s = ensure_dist(c))) return s;
  c->dist_direct = mivi_comm_route(c) == 3 && p2p_direct_ok(c, params);
  if (c->dist_direct) {   // the partial kernels store into the owners' staging areas (group 0, vector 0 of the exchange launched below)
    if ((s = ensure_work(c, c->cfg.n_mc))) return s;
    OutArgs o = final_out(c, nullptr, nullptr);
    o.partials = c->dist_P.p;
    o.partials_mode = 1;
    o.scalars_off = mivi_partials_len(c) - 2;
    o.p2p_direct = c->p2p_direct.p;
    o.p2p_gi = 0; o.p2p_v = 0;
    if ((s = run_estimate(c, params, rng_of(c, idx), c->cfg.n_mc, 1, o))) return s;
  } else if ((s = mivi_estimate_partials(c, params, idx, c->dist_P.p))) {
    return s;
  }
  return dist_collective(c, params, c->dist_P.p, value, grad);
}

// ---------------------------------------------------------------------------------------------
// Sharded estimates in batches: the exchange of estimate t overlapped with the kernels of estimate t + 1
// ---------------------------------------------------------------------------------------------
// Estimates at fixed parameters are independent (what mivi_estimate_gradient_n serves on one GPU), so the exchange + finalisation of
// estimate t (comm_stream) runs UNDER the partial kernels of estimate t + 1 (the context's stream): partial vectors are double
// buffered, estimate t + 2 waits for the exchange of t to release its buffer.  One hipGraph with two branches per estimate; if the
// capture is refused (a collective that cannot be captured) the same sequence is issued eagerly with events.
//   mode 0 pipelined | 1 serial {partials -> exchange} on one stream | 2 partials only | 3 exchange only (on the last partial vector)
// mode 4 with LANE-BATCHED compute (second-generation full-rank kernels, no STL solve): four contexts compute four consecutive estimates
// with ONE product launch and ONE VJP launch (blockIdx.y = lane, each lane's packed partials into its ring slot), ONE hand-over per four
// estimates -- the exchange kernel serves them as one group (kernels_p2p.hip).
static mivi_status_t dist_sequence_lanes(mivi_ctx *c, const void *params, bool counter_idx, uint64_t idx0, int count) {
  constexpr int E = kGroup;
  mivi_status_t s = MIVI_OK;
  void *ringP[kRing] = {c->dist_P.p, c->dist_P2.p, c->dist_ring[0].p, c->dist_ring[1].p, c->dist_ring[2].p, c->dist_ring[3].p, c->dist_ring[4].p, c->dist_ring[5].p};
  mivi_ctx *ctxs[E];
  hipStream_t kept[E];
  ctxs[0] = c;
  for (int l = 1; l < E; ++l) ctxs[l] = c->kids[l - 1];
  LaneSink *sink = lane_sinks_alloc(E);
  EpsSink *esink = eps_sink_alloc();
  const bool dense = c->target == TGT_DENSE_GAUSS;
  for (int l = 0; l < E; ++l) { kept[l] = ctxs[l]->stream; ctxs[l]->stream = c->stream; ctxs[l]->lane_sink = sink; ctxs[l]->lane_id = l; ctxs[l]->eps_sink = esink; }
  unsigned *w = (unsigned *)c->p2p_ctr.p;
  for (int s0 = 0; s0 < count && s == MIVI_OK; s0 += E) {
    const int L = count - s0 < E ? count - s0 : E;
    eps_sink_reset(esink);
    for (int l = 0; l < L && s == MIVI_OK; ++l) {
      const int i = s0 + l;
      mivi_ctx *k = ctxs[l];
      lane_sink_reset(sink, l);
      RngArgs r = rng_of(k, counter_idx ? (uint64_t)i : idx0 + (uint64_t)i);
      if (counter_idx) r.idx_ptr = (const uint64_t *)c->d_idx.p;
      OutArgs o = final_out(k, nullptr, nullptr);
      o.partials = ringP[i % kRing];
      o.partials_mode = 1;
      o.scalars_off = mivi_partials_len(c) - 2;
      if (c->dist_direct) { o.p2p_direct = c->p2p_direct.p; o.p2p_gi = i / kGroup; o.p2p_v = i % kGroup; }
      if ((s = run_estimate(k, params, r, k->cfg.n_mc, 1, o))) { c->err = k->err; break; }
      if (lane_sink_counts(sink, l) != (dense ? 2 : 1) * 16 + 1) s = fail(c, MIVI_ERR_HIP, "lane-batched sharded estimates: an estimate did not take the two-kernel route");
    }
    if (s == MIVI_OK) launch_lanes_eps(c, esink, L);
    if (s == MIVI_OK && !(launch_lanes_prod(c, sink, L, 0) && (!dense || launch_lanes_prod(c, sink, L, 1)) && launch_lanes_vjp(c, sink, L)))
      s = fail(c, MIVI_ERR_HIP, "lane-batched sharded estimates: the lanes' launches do not match");
    if (s) break;
    // announce the group's partial vectors; hold the chain until the exchange has read the ring slots the NEXT group overwrites
    const unsigned *fr[E];
    unsigned fmin[E];
    int nf = 0;
    for (int l = 0; l < E; ++l) {
      const int nx = s0 + E + l;
      if (nx >= count) break;
      const int prev_users = nx / kRing;
      if (prev_users >= 1) { fr[nf] = w + 80 + nx % kRing; fmin[nf] = (unsigned)prev_users * (unsigned)(c->p2p_G + (c->dist_direct ? 1 : 0)); ++nf; }   // (direct: the value workgroup releases a slot too)
    }
    launch_p2p_handover4(c, w + 64, (unsigned)(s0 + L), fr, fmin, nf);
  }
  for (int l = 0; l < E; ++l) {
    ctxs[l]->lane_sink = nullptr;
    ctxs[l]->eps_sink = nullptr;
    ctxs[l]->stream = kept[l];
    ctxs[l]->cur = 0;
    ctxs[l]->pre_valid = false;
  }
  lane_sinks_free(sink);
  eps_sink_free(esink);
  return s;
}

static mivi_status_t dist_sequence(mivi_ctx *c, const void *params, bool counter_idx, uint64_t idx0, int count, void *value, void *grad, int mode) {
  if (mode == 4 && c->dist_lane4) return dist_sequence_lanes(c, params, counter_idx, idx0, count);
  mivi_status_t s = MIVI_OK;
  hipStream_t main = c->stream;
  if (mode == 3 && c->dist_direct) {   // exchange-only (a measurement leg): both parities of the staging areas hold a complete partial vector
    for (int g2 = 0; g2 < 2 && s == MIVI_OK; ++g2) {
      OutArgs o = final_out(c, nullptr, nullptr);
      o.partials = c->dist_P.p;
      o.partials_mode = 1;
      o.scalars_off = mivi_partials_len(c) - 2;
      o.p2p_direct = c->p2p_direct.p;
      o.p2p_gi = g2; o.p2p_v = 0;
      s = run_estimate(c, params, rng_of(c, idx0), c->cfg.n_mc, 1, o);
    }
  }
  for (int i = 0; i < count && s == MIVI_OK; ++i) {
    const int par = i & 1;
    void *ringP[kRing] = {c->dist_P.p, c->dist_P2.p, c->dist_ring[0].p, c->dist_ring[1].p, c->dist_ring[2].p, c->dist_ring[3].p, c->dist_ring[4].p, c->dist_ring[5].p};
    void *P = mode == 4 ? ringP[i % kRing] : (par ? c->dist_P2.p : c->dist_P.p);
    if (mode == 0 && i >= 2) HIPCHK(c, hipStreamWaitEvent(main, c->ev_comm[par], 0));   // the exchange of i - 2 has released this partial buffer
    if (mode != 3) {
      RngArgs r = rng_of(c, counter_idx ? (uint64_t)i : idx0 + (uint64_t)i);
      if (counter_idx) r.idx_ptr = (const uint64_t *)c->d_idx.p;
      OutArgs o = final_out(c, nullptr, nullptr);
      o.partials = P;
      o.partials_mode = 1;
      o.scalars_off = mivi_partials_len(c) - 2;
      if (c->dist_direct) {   // mode 4: the persistent exchange serves groups of kGroup estimates; modes 1 / 2: one exchange launch per estimate
        o.p2p_direct = c->p2p_direct.p;
        o.p2p_gi = mode == 4 ? i / kGroup : 0;
        o.p2p_v = mode == 4 ? i % kGroup : 0;
      }
      if ((s = run_estimate(c, params, r, c->cfg.n_mc, 1, o))) break;
    }
    if (mode == 2) continue;
    if (mode == 4) {   // peer-to-peer pipeline, compute chain: announce partial vector i, then wait until the exchange has read the ring slot estimate i + 1 overwrites
      unsigned *w = (unsigned *)c->p2p_ctr.p;
      const int slot = (i + 1) % kRing, prev_users = (i + 1) / kRing;
      // (folding this one-thread launch into the next estimate's product kernel as an extra workgroup was tried: the 8 us it takes from
      //  dispatch to completion beside the persistent exchange kernels moved into that kernel -- 9 + 8 -> 21.5 us --, the step stayed at 31 us)
      launch_p2p_handover(c, w + 64, (unsigned)i + 1u, prev_users >= 1 ? w + 80 + slot : nullptr,
                          (unsigned)prev_users * (unsigned)(c->p2p_G + (c->dist_direct ? 1 : 0)));
      continue;
    }
    if (mode == 0) {
      HIPCHK(c, hipEventRecord(c->ev_part[par], main));
      HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_part[par], 0));
      c->stream = c->comm_stream;
      s = dist_collective(c, params, P, value, grad);
      c->stream = main;
      if (s) break;
      HIPCHK(c, hipEventRecord(c->ev_comm[par], c->comm_stream));
    } else {
      s = dist_collective(c, params, mode == 3 ? c->dist_P.p : P, value, grad);
    }
  }
  if (s == MIVI_OK && mode == 0) {   // join: the batch is complete when its last two exchanges are
    if (count >= 2) HIPCHK(c, hipStreamWaitEvent(main, c->ev_comm[(count - 2) & 1], 0));
    HIPCHK(c, hipStreamWaitEvent(main, c->ev_comm[(count - 1) & 1], 0));
  }
  return s;
}

// the one collective of the engine's sharded batches: the lanes' partial vectors summed over the ranks, in place, on the context's stream
mivi_status_t dist_allreduce_f32(mivi_ctx *c, void *buf, size_t count, hipStream_t stream) {
  if (!c->comm) return MIVI_OK;   // (one rank without a communicator: the sum over the ranks is the vector itself)
  RcclApi *r = rccl();
  if (!r || !r->AllReduce) return fail(c, MIVI_ERR_UNSUPPORTED, "this librccl exports no ncclAllReduce");
  if (r->AllReduce(buf, buf, count, ncclFloat, ncclSum, (ncclComm_t)c->comm, stream) != ncclSuccess) return fail(c, MIVI_ERR_HIP, "ncclAllReduce failed");
  return MIVI_OK;
}
mivi_status_t dist_comm_stream(mivi_ctx *c) {
  if (c->fb_comm_stream) return MIVI_OK;
  HIPCHK(c, hipStreamCreateWithFlags(&c->fb_comm_stream, hipStreamNonBlocking));
  for (int k = 0; k < 2; ++k) {
    HIPCHK(c, hipEventCreateWithFlags(&c->fb_ev_part[k], hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&c->fb_ev_comm[k], hipEventDisableTiming));
  }
  return MIVI_OK;
}

static mivi_status_t dist_batch(mivi_ctx *c, const void *params, uint64_t idx0, int count, void *value, void *grad, int mode) {
  if (!graph_capturable(c)) return fail(c, MIVI_ERR_UNSUPPORTED, "batched sharded estimates need a device-resident built-in target");
  if (c->idx_src) return fail(c, MIVI_ERR_UNSUPPORTED, "an index source is set (mivi_set_index_source): batched calls keep their own device counter");
  if (c->p2p_on && !c->comm) { c->comm_world = c->p2p_world; c->comm_rank = c->p2p_rank; }
  mivi_status_t s;
  // Round 6: on a batch-engine shape (full-rank f32, d and n_mc multiples of 128, Gaussian target) and an RCCL route -- or one rank without
  // peer-to-peer areas -- the batch runs on the ENGINE: draws, product, VJP for up to 80 estimates per step as on one GPU, the lanes' partial
  // vectors summed by ONE all-reduce per step (20 lanes x 2.4 MB: one large collective instead of twenty 2.1 MB ones), one finalisation
  // launch.  The peer-to-peer route keeps the four-estimate kernels its persistent exchange kernel is built around.
  if (mode == 0 && count >= 2 && mivi_comm_route(c) != 3 && (c->comm || c->comm_world <= 1) && fb_dist_route(c, params, grad)) {
    if ((s = dist_check(c))) return s;
    return fb_batch_dist(c, params, idx0, count, value, grad);
  }
  if ((s = dist_check(c)) || (s = ensure_work(c, c->cfg.n_mc))) return s;
  prepare_tables(c, c->cfg.n_mc);
  if ((s = reserve_target(c, c->cfg.n_mc)) || (s = ensure_dist(c))) return s;
  if (c->cfg.family == MIVI_FULLRANK && lds_path_shape_ok(c, c->cfg.n_mc) && !lds_prepare(c, c->cfg.n_mc)) return fail(c, MIVI_ERR_HIP, "full-rank work lists: allocation failed");
  if (!c->comm_stream) {
    // (plain non-blocking streams: a HIGH-priority stream starved the compute chain it was supposed to run beside -- its spinning
    //  kernel was scheduled first and the normal-priority graph never progressed; found on the GPU)
    // The persistent exchange kernels run BESIDE the compute chain on these streams.  Non-blocking: a blocking stream (what
    // hipExtStreamCreateWithCUMask creates) synchronises with the null stream, so a context living on the null stream deadlocked
    // against its own exchange kernel; a HIGH-priority stream starved the compute chain (both found on the GPU).
    // They need hardware queues of their own (HIP maps streams onto a small pool, GPU_MAX_HW_QUEUES, and two streams on one queue
    // serialise: the bounded hand-over waits then expire): a stream created with a CU mask carries the mask in its queue and gets one
    // -- all CUs enabled = no restriction.  Only for contexts on a real stream (see above).
    uint32_t mask[16];
    for (int k = 0; k < 16; ++k) mask[k] = 0xFFFFFFFFu;
    hipStream_t *cs[2] = {&c->comm_stream, &c->comm_stream2};
    for (int k = 0; k < 2; ++k) {
      if (c->stream == nullptr || hipExtStreamCreateWithCUMask(cs[k], 16, mask) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(c, hipStreamCreateWithFlags(cs[k], hipStreamNonBlocking));
      }
    }
    for (int k = 0; k < 2; ++k) {
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_part[k], hipEventDisableTiming));
      HIPCHK(c, hipEventCreateWithFlags(&c->ev_comm[k], hipEventDisableTiming));
    }
  }
  GraphCache &g = c->graph;
  const int route = mivi_comm_route(c);
  // Peer-to-peer route, pipelined: the exchange is ONE persistent kernel on comm_stream for the whole batch (kernels_p2p.hip), the compute
  // chain is a single-stream graph of {partial kernels, hand-over} per estimate; the two talk through two device words.
  {
    const bool direct = route == 3 && p2p_direct_ok(c, params);
    if (c->dist_direct != direct) { invalidate_graph(c); c->dist_direct = direct; }
  }
  bool p2p_pipe = mode == 0 && route == 3;
  if (p2p_pipe && c->p2p_pipe_state < 0) { p2p_pipe = false; mode = 1; }   // (its kernels did not run beside the compute chain on this context: serial steps)
  if (p2p_pipe) mode = 4;
  {   // lane-batched compute chain for the pipelined batches (see dist_sequence_lanes)
    static const bool no_lanes = getenv("MIVI_LANE_BATCH") && atoi(getenv("MIVI_LANE_BATCH")) == 0;
    const bool stl_ent = c->cfg.entropy == MIVI_ENT_STL || c->cfg.entropy == MIVI_ENT_STL_ZERO_GRAD;
    OutArgs on = final_out(c, nullptr, nullptr);
    on.partials = c->dist_P.p;
    on.partials_mode = 1;
    const bool lane4 = mode == 4 && !no_lanes && !stl_ent && !c->dbg && c->cfg.family == MIVI_FULLRANK && lds_route(c, params, c->cfg.n_mc, 1, on) &&
                       lds_use_prod32(c, c->cfg.n_mc) && lds_bf16x3() && count >= kGroup;
    const int stride = lane4 ? kGroup : 1;
    if (c->dist_lane4 != lane4 || c->idx_stride != stride) { invalidate_graph(c); c->dist_lane4 = lane4; c->idx_stride = stride; }
    if (lane4) {
      if ((s = ensure_kids(c, kGroup))) return s;
      for (int j = 0; j < kGroup - 1; ++j) {
        mivi_ctx *k = c->kids[j];
        if ((s = sync_kid(c, k, kGroup)) || (s = ensure_work(k, k->cfg.n_mc))) { c->err = k->err; return s; }
        prepare_tables(k, k->cfg.n_mc);
        if (!lds_prepare(k, k->cfg.n_mc)) return fail(c, MIVI_ERR_HIP, "full-rank work lists: allocation failed");
      }
    }
  }
  const int kind = 20 + mode;
  bool &capture_refused = c->dist_capture_refused;   // (a collective library that cannot be captured: do not retry on every call of THIS context)
  if (!(g.exec && g.kind == kind && g.count == count && g.params == params && g.value == value && g.grad == grad && g.p0 == (double)route) && !capture_refused) {
    invalidate_graph(c);
    if (c->dist_lane4) {
      for (int j = 0; j < kGroup - 1; ++j) {
        c->kids[j]->kid_gen = c->target_gen;
        HIPCHK(c, hipStreamSynchronize(c->kids[j]->stream));   // (their table uploads, before the capture)
      }
    }
    hipGraph_t graph = nullptr;
    hipStream_t saved;
    if ((s = begin_capture(c, &saved))) return s;
    s = dist_sequence(c, params, true, 0, count, value, grad, mode);
    if (s == MIVI_OK && mode != 3) hipLaunchKernelGGL(k_bump_u64, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, (uint64_t)count);
    c->cur = 0;
    c->pre_valid = false;
    hipError_t e = end_capture(c, saved, &graph);
    if (s == MIVI_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (s != MIVI_OK || e != hipSuccess) {
      (void)hipGetLastError();
      g = GraphCache{};
      capture_refused = true;
      (void)hipStreamSynchronize(c->comm_stream);
    } else {
      g.kind = kind; g.count = count; g.params = params; g.value = value; g.grad = grad; g.p0 = (double)route;
    }
  }
  auto p2p_front = [&]() -> mivi_status_t {   // hand-over words reset, then the persistent exchange kernels (one per lane) on their own streams
    unsigned *w = (unsigned *)c->p2p_ctr.p;
    HIPCHK(c, hipMemsetAsync(w + 64, 0, 128, c->stream));
    HIPCHK(c, hipEventRecord(c->ev_part[0], c->stream));
    hipStream_t main = c->stream;
    const void *ringP[kRing] = {c->dist_P.p, c->dist_P2.p, c->dist_ring[0].p, c->dist_ring[1].p, c->dist_ring[2].p, c->dist_ring[3].p, c->dist_ring[4].p, c->dist_ring[5].p};
    // ONE persistent exchange kernel (measured on one GPU: 21 us per estimate against 31 with two of them serving alternate estimates -- a
    // second resident kernel costs the compute chain more than its overlap wins)
    const int lanes = 1;
    for (int ln = 0; ln < lanes; ++ln) {
      hipStream_t cs = ln ? c->comm_stream2 : c->comm_stream;
      HIPCHK(c, hipStreamWaitEvent(cs, c->ev_part[0], 0));
      c->stream = cs;
      launch_p2p_exchange(c, params, ringP, kRing, value, grad, 7, ln, lanes, count, w + 64, w + 80, c->dist_direct);
      c->stream = main;
      HIPCHK(c, hipEventRecord(c->ev_comm[ln], cs));
    }
    return MIVI_OK;
  };
  auto p2p_back = [&]() -> mivi_status_t {
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_comm[0], 0));
    return MIVI_OK;
  };
  if (g.exec && g.kind == kind && g.count == count && g.params == params && g.value == value && g.grad == grad && g.p0 == (double)route) {
    if (!(c->d_idx_valid && c->d_idx_expect == idx0))
      hipLaunchKernelGGL(k_set_u64x2, dim3(1), dim3(1), 0, c->stream, (uint64_t *)c->d_idx.p, idx0, 0ull, 1);
    if (p2p_pipe && (s = p2p_front())) return s;
    const hipError_t ge = hipGraphLaunch(g.exec, c->stream);
    if (p2p_pipe && (s = p2p_back())) return s;   // (also behind a failed launch: the exchange kernel's waits are bounded, the caller's stream joins it)
    HIPCHK(c, ge);
    c->d_idx_valid = mode != 3;
    c->d_idx_expect = idx0 + (uint64_t)count;
    return MIVI_OK;
  }
  // eager: the same sequence with by-value indices
  c->pre_valid = false;
  if (p2p_pipe && (s = p2p_front())) return s;
  s = dist_sequence(c, params, false, idx0, count, value, grad, mode);
  if (p2p_pipe) {   // join the exchange stream whatever happened: after a failed sequence its kernel gives up at its bounded waits (status bit 8)
    const mivi_status_t sb = p2p_back();
    if (s == MIVI_OK) s = sb;
  }
  c->cur = 0;
  c->pre_valid = false;
  return s;
}

mivi_status_t mivi_estimate_gradient_dist_n(mivi_ctx_t *c, const void *params, uint64_t idx0, int32_t count, void *value, void *grad) {
  if (!c || !params || !value || !grad || count <= 0) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  return dist_batch(c, params, idx0, count, value, grad, 0);
}

// tests: the phases of the peer-to-peer exchange one launch at a time (several ranks of ONE process driven from one host thread)
mivi_status_t mivi_p2p_exchange(mivi_ctx_t *c, const void *params, const void *partials, void *value, void *grad, int32_t phases) {
  if (!c || !params || !value || !grad || phases < 1 || phases > 7) return MIVI_ERR_BAD_ARG;
  if (!c->p2p_on) return fail(c, MIVI_ERR_BAD_ARG, "no peer-to-peer exchange buffers attached");
  (void)hipSetDevice(c->cfg.device);
  const void *Ps[1] = {partials};   // NULL: direct mode -- mivi_p2p_partials_direct stored the vector into the owners' staging areas
  launch_p2p_exchange(c, params, Ps, 1, value, grad, phases, 0, 1, 1, nullptr, nullptr, partials == nullptr);
  HIPCHK(c, hipGetLastError());
  return MIVI_OK;
}

// Tests: the partial kernels of ONE estimate in direct mode -- every entry of this rank's partial vector goes straight into its owner's
// staging area, as vector 0 of the exchange launched next (mivi_p2p_exchange with partials = NULL).  Second-generation full-rank f32 route.
mivi_status_t mivi_p2p_partials_direct(mivi_ctx_t *c, const void *params, uint64_t idx) {
  if (!c || !params) return MIVI_ERR_BAD_ARG;
  if (!c->p2p_on) return fail(c, MIVI_ERR_BAD_ARG, "no peer-to-peer exchange buffers attached");
  (void)hipSetDevice(c->cfg.device);
  if (c->target == TGT_NONE) return fail(c, MIVI_ERR_NO_TARGET, "no target set");
  mivi_status_t s;
  if ((s = ensure_work(c, c->cfg.n_mc)) || (s = ensure_dist(c))) return s;
  OutArgs o = final_out(c, nullptr, nullptr);
  o.partials = c->dist_P.p;
  o.partials_mode = 1;
  o.scalars_off = mivi_partials_len(c) - 2;
  if (!c->p2p_direct.p || c->cfg.family != MIVI_FULLRANK || c->cfg.dtype != MIVI_F32 || !lds_route(c, params, c->cfg.n_mc, 1, o))
    return fail(c, MIVI_ERR_UNSUPPORTED, "direct staging serves the second-generation full-rank f32 kernels");
  o.p2p_direct = c->p2p_direct.p;
  o.p2p_gi = 0; o.p2p_v = 0;
  return run_estimate(c, params, rng_of(c, idx), c->cfg.n_mc, 1, o);
}

// us per estimate of the sharded step and of its pieces, every rank calling collectively: out[0] partial kernels, out[1] exchange +
// finalisation, out[2] serial step (one stream), out[3] pipelined step (mivi_estimate_gradient_dist_n).  hipEvents on the context's
// stream around ONE graph replay of `reps` estimates each (after a warm replay).
mivi_status_t mivi_profile_dist(mivi_ctx_t *c, const void *params, int32_t reps, double *us_out) {
  if (!c || !params || reps <= 0 || !us_out) return MIVI_ERR_BAD_ARG;
  (void)hipSetDevice(c->cfg.device);
  char *o = (char *)c->tmp_out.p;
  hipEvent_t e0, e1;
  HIPCHK(c, hipEventCreate(&e0));
  HIPCHK(c, hipEventCreate(&e1));
  mivi_status_t s = MIVI_OK;
  const int order[4] = {2, 3, 1, 0};   // partials first: the exchange-only leg works on the partial vector they leave
  for (int k = 0; k < 4 && s == MIVI_OK; ++k) {
    const int mode = order[k];
    if ((s = dist_batch(c, params, 1000, reps, o, o + 16, mode))) break;   // warm (captures)
    HIPCHK(c, hipEventRecord(e0, c->stream));
    if ((s = dist_batch(c, params, 1000 + reps, reps, o, o + 16, mode))) break;
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, e0, e1));
    us_out[mode == 2 ? 0 : (mode == 3 ? 1 : (mode == 1 ? 2 : 3))] = (double)ms * 1e3 / reps;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  invalidate_graph(c);
  return s;
}

