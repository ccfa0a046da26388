// abi_group.hip -- multi-rank: RCCL communicators, the one-hop peer exchange's set-up over hipIpc, single-process groups.
#include "ctx.hpp"

RSRL_API_BEGIN

int rsrl_hip_comm_unique_id(uint8_t* id_bytes) {
    if (!id_bytes) return fail(RSRL_HIP_EINVAL, "null argument");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes in the ABI");
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    memcpy(id_bytes, &id, sizeof(id));
    return RSRL_HIP_OK;
}
int rsrl_hip_comm_init(rsrl_hip_ctx* c, const uint8_t* id_bytes, int world_size, int rank) {
    CHECK_CTX(c); FLUSH(c);
    if (!id_bytes || world_size < 1 || rank < 0 || rank >= world_size) return fail(RSRL_HIP_EINVAL, "bad communicator arguments");
    if (c->comm) return fail(RSRL_HIP_ESTATE, "communicator already initialised");
    if (c->cfg.weight_mode != RSRL_W_SHARED) return fail(RSRL_HIP_ESTATE, "per-env weights need no collective: shard by env_offset instead");
    HIP_TRY(hipSetDevice(c->cfg.device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    if (c->multi) return fail(RSRL_HIP_ESTATE, "an exchange is already attached");
    if (c->cfg.exchange == RSRL_EXCHANGE_PEER) return fail(RSRL_HIP_ESTATE, "this ctx was configured for the peer exchange: use rsrl_hip_peer_export / _connect");
    // (AUTO: attaching a communicator decides -- but only once it IS attached: a failed attach leaves the ctx as configured, so that a host
    // can still fall back to the other exchange)
    {
        const ncclResult_t nr = ncclCommInitRank(&c->comm, world_size, id, rank);
        if (nr != ncclSuccess) { c->comm = nullptr; return fail(RSRL_HIP_ERCCL, "ncclCommInitRank: %s", ncclGetErrorString(nr)); }
    }
    // warm-up: RCCL sets its connections up lazily, at the first collective -- which must not be the one inside the step graph's
    // stream capture.  dW is zero between operations, so all-reducing it leaves it zero; every rank makes this call (comm_init is
    // collective by nature).
    {
        const ncclResult_t nr = ncclAllReduce(c->dW, c->dW, c->dw_elems, ncclFloat, ncclSum, c->comm, c->stream);
        const hipError_t he = nr == ncclSuccess ? hipStreamSynchronize(c->stream) : hipSuccess;
        if (nr != ncclSuccess || he != hipSuccess) {
            (void)ncclCommAbort(c->comm); c->comm = nullptr; (void)hipGetLastError();
            return nr != ncclSuccess ? fail(RSRL_HIP_ERCCL, "warm-up all-reduce: %s", ncclGetErrorString(nr))
                                     : fail(RSRL_HIP_EHIP, "warm-up all-reduce: %s", hipGetErrorString(he));
        }
    }
    c->cfg.exchange = RSRL_EXCHANGE_RCCL;
    c->world_size = world_size; c->rank = rank; c->multi = true;
    return RSRL_HIP_OK;
}

int rsrl_hip_comm_info(rsrl_hip_ctx* c, int* world_size, int* rank, int* exchange) {
    CHECK_CTX(c);
    int w = 1, r = 0;
    if (c->comm) { NCCL_TRY(ncclCommCount(c->comm, &w)); NCCL_TRY(ncclCommUserRank(c->comm, &r)); }       // what RCCL itself reports
    else if (c->multi) { w = c->world_size; r = c->rank; }
    if (world_size) *world_size = w;
    if (rank) *rank = r;
    if (exchange) *exchange = !c->multi ? -1 : c->cfg.exchange;
    return RSRL_HIP_OK;
}

// ---- RSRL_EXCHANGE_PEER set-up: export this rank's receive buffer, connect to everybody's -----------------------------
// what a rank tells the others about itself: where its receive buffer is -- and what rsrl_hip_peer_connect needs to decide, the same way
// on every rank, whether the group runs the persistent kernel: the rank's grid (rows), what its device admits (budgets), which
// physical device that is (ranks of one node may share one), and whether it could run the kernel at all (flags bit 0)
struct PeerBlob { uint32_t magic; int32_t pid; uint64_t ptr; uint64_t bytes; int32_t world; uint32_t sh_rows; hipIpcMemHandle_t h;
                  uint64_t dev_id; uint32_t budget_shared; uint32_t flags; };
// identity of the physical device behind a ctx, the same in every process of the node (ordinals are not: HIP_VISIBLE_DEVICES)
static uint64_t device_identity(int device) {
    int dom = 0, bus = 0, dev = 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainID, device) != hipSuccess) { (void)hipGetLastError(); dom = 0; }
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) != hipSuccess) { (void)hipGetLastError(); bus = device; }
    if (hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    return ((uint64_t)(uint32_t)dom << 32) | ((uint64_t)(uint32_t)(bus & 0xffff) << 16) | (uint64_t)(uint32_t)(dev & 0xffff) | (1ull << 63);
}
static_assert(sizeof(PeerBlob) <= RSRL_HIP_PEER_HANDLE_BYTES, "peer handle blob must fit the ABI slot");
int rsrl_hip_peer_export(rsrl_hip_ctx* c, int world_size, uint8_t* handle_out) {
    CHECK_CTX(c); FLUSH(c);
    if (!handle_out || world_size < 1 || world_size > 64) return fail(RSRL_HIP_EINVAL, "bad peer arguments");
    if (c->cfg.weight_mode != RSRL_W_SHARED) return fail(RSRL_HIP_ESTATE, "per-env weights need no exchange: shard by env_offset instead");
    if (c->cfg.exchange == RSRL_EXCHANGE_RCCL) return fail(RSRL_HIP_ESTATE, "this ctx was configured for the RCCL exchange: use rsrl_hip_comm_init");
    if (c->multi || c->peer_recv) return fail(RSRL_HIP_ESTATE, "an exchange is already attached");
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->peer_old_bytes = sizeof(uint2) * 2 * (size_t)world_size * c->dw_elems;
    // second region: the hop-2 buffer of the persistent kernel, [2 (parity)][world][A*F rounded up to even] granules
    c->peer_recv_bytes = c->peer_old_bytes + sizeof(unsigned long long) * 2 * (size_t)world_size * (((size_t)c->dw_elems + 1) / 2 * 2);
    // fine-grained (uncached across agents) memory, as RCCL uses for its own flag/buffer exchange; RSRL_PEER_COARSE=1 falls
    // back to a plain allocation (same-device peers only need the system-scope accesses the kernels already use)
    hipError_t e = getenv("RSRL_PEER_COARSE") ? hipErrorNotSupported
                                             : hipExtMallocWithFlags((void**)&c->peer_recv, c->peer_recv_bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); c->peer_recv = nullptr; e = hipMalloc((void**)&c->peer_recv, c->peer_recv_bytes); }
    PeerBlob b; memset(&b, 0, sizeof(b));
    if (e == hipSuccess) e = hipMemsetAsync(c->peer_recv, 0, c->peer_recv_bytes, c->stream);      // tag 0 never matches a batch-step (tags start at 1)
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&b.h, c->peer_recv);
    if (e != hipSuccess) {
        // (AUTO: exporting a receive buffer decides -- but only once it HAS been exported: a failed export leaves the ctx as configured and
        // without a receive buffer, so that a host can still attach the RCCL exchange)
        (void)hipGetLastError();
        if (c->peer_recv) { (void)hipFree(c->peer_recv); c->peer_recv = nullptr; }
        c->peer_recv_bytes = 0; c->peer_old_bytes = 0;
        return fail(e == hipErrorOutOfMemory ? RSRL_HIP_ENOMEM : RSRL_HIP_EHIP, "exporting the receive buffer: %s", hipGetErrorString(e));
    }
    c->cfg.exchange = RSRL_EXCHANGE_PEER;
    c->peer_world = world_size;
    b.magic = 0x52504552u; b.pid = (int32_t)getpid(); b.ptr = (uint64_t)(uintptr_t)c->peer_recv; b.bytes = c->peer_recv_bytes; b.world = world_size;
    b.sh_rows = c->sh_rows; b.dev_id = device_identity(c->cfg.device);
    b.budget_shared = persist_budget_shared(c);
    b.flags = persist_capable(c) ? 1u : 0u;                            // (RSRL_NO_PERSIST in this rank's environment included: it travels to the others)
    memset(handle_out, 0, RSRL_HIP_PEER_HANDLE_BYTES);
    memcpy(handle_out, &b, sizeof(b));
    return RSRL_HIP_OK;
}
/* identity of the physical device behind ordinal `device` -- PCI domain : bus : device, the same number in every process of the node whatever
 * HIP_VISIBLE_DEVICES each of them runs under (ordinals are not) */
int rsrl_hip_device_identity(int device, uint64_t* identity_out) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (!identity_out) return fail(RSRL_HIP_EINVAL, "null argument");
    if (device < 0 || device >= n) return fail(RSRL_HIP_EINVAL, "device out of range (%d devices)", n);
    *identity_out = device_identity(device);
    return RSRL_HIP_OK;
}
int rsrl_hip_can_access_peer(int device, int peer_device) {
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n || peer_device < 0 || peer_device >= n) return fail(RSRL_HIP_EINVAL, "device out of range (%d devices)", n);
    if (device == peer_device) return 1;
    int can = 0;
    HIP_TRY(hipDeviceCanAccessPeer(&can, device, peer_device));
    return can ? 1 : 0;
}
int rsrl_hip_peer_connect(rsrl_hip_ctx* c, const uint8_t* handles, int world_size, int rank) {
    CHECK_CTX(c); FLUSH(c);
    if (!handles || world_size < 1 || rank < 0 || rank >= world_size) return fail(RSRL_HIP_EINVAL, "bad peer arguments");
    if (!c->peer_recv || c->peer_world != world_size) return fail(RSRL_HIP_ESTATE, "call rsrl_hip_peer_export(world_size) first");
    if (c->multi) return fail(RSRL_HIP_ESTATE, "an exchange is already attached");
    HIP_TRY(hipSetDevice(c->cfg.device));
    c->peer_ptrs.assign((size_t)world_size, nullptr); c->peer_opened.assign((size_t)world_size, 0);
    for (int r = 0; r < world_size; ++r) {
        PeerBlob b; memcpy(&b, handles + (size_t)r * RSRL_HIP_PEER_HANDLE_BYTES, sizeof(b));
        if (b.magic != 0x52504552u || b.world != world_size || b.bytes != c->peer_recv_bytes)
            return fail(RSRL_HIP_EINVAL, "peer handle %d does not describe a matching receive buffer", r);
        if (r == rank) {
            if ((uint64_t)(uintptr_t)c->peer_recv != b.ptr || b.pid != (int32_t)getpid()) return fail(RSRL_HIP_EINVAL, "handle %d is not this ctx's own export", r);
            c->peer_ptrs[r] = c->peer_recv;
        } else if (b.pid == (int32_t)getpid()) {
            // a ctx of this very process (several ranks driven by one host process): its pointer is valid here -- once this
            // ctx's device may access the memory of the device it lives on
            c->peer_ptrs[r] = (void*)(uintptr_t)b.ptr;
            hipPointerAttribute_t attr;
            HIP_TRY(hipPointerGetAttributes(&attr, c->peer_ptrs[r]));
            if (attr.device != c->cfg.device) {
                int can = 0;
                HIP_TRY(hipDeviceCanAccessPeer(&can, c->cfg.device, attr.device));
                if (!can) return fail(RSRL_HIP_EINVAL, "device %d cannot access the memory of device %d (peer rank %d): use the RCCL exchange", c->cfg.device, attr.device, r);
                const hipError_t pe = hipDeviceEnablePeerAccess(attr.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled)
                    return fail(RSRL_HIP_EHIP, "hipDeviceEnablePeerAccess(%d) from device %d: %s", attr.device, c->cfg.device, hipGetErrorString(pe));
                (void)hipGetLastError();
            }
        } else {
            HIP_TRY(hipIpcOpenMemHandle(&c->peer_ptrs[r], b.h, hipIpcMemLazyEnablePeerAccess));
            c->peer_opened[r] = 1;
        }
    }
    HIP_TRY(hipMalloc((void**)&c->d_peer_ptrs, sizeof(void*) * (size_t)world_size));
    HIP_TRY(hipMemcpy(c->d_peer_ptrs, c->peer_ptrs.data(), sizeof(void*) * (size_t)world_size, hipMemcpyHostToDevice));
    {   // the persistent kernel's hop-2 buffers: the second region of every rank's receive buffer
        std::vector<unsigned long long*> bp((size_t)world_size);
        for (int r = 0; r < world_size; ++r) bp[(size_t)r] = reinterpret_cast<unsigned long long*>(static_cast<char*>(c->peer_ptrs[(size_t)r]) + c->peer_old_bytes);
        if (c->px_B && c->px_B_owned) { HIP_TRY(hipFree(c->px_B)); }
        c->px_B = bp[(size_t)rank]; c->px_B_owned = false;
        if (c->d_px_Bptrs) { HIP_TRY(hipFree(c->d_px_Bptrs)); c->d_px_Bptrs = nullptr; }
        HIP_TRY(hipMalloc((void**)&c->d_px_Bptrs, sizeof(void*) * (size_t)world_size));
        HIP_TRY(hipMemcpy(c->d_px_Bptrs, bp.data(), sizeof(void*) * (size_t)world_size, hipMemcpyHostToDevice));
        c->px_seq = 0;                                      // a fresh (cleared) hop-2 buffer: the sequence restarts, on every rank alike
        if (c->px_A) { HIP_TRY(hipFree(c->px_A)); c->px_A = nullptr; }
    }
    {   // (2) the persistent kernel or the per-step kernels: ONE decision for the whole group, computed by every rank from the same
        // handles.  Persistent iff every rank could run it alone AND, on every device that hosts several ranks, the sum of their grids
        // fits the smallest budget any of them reported for it.
        std::vector<PeerBlob> bl((size_t)world_size);
        for (int r = 0; r < world_size; ++r) memcpy(&bl[(size_t)r], handles + (size_t)r * RSRL_HIP_PEER_HANDLE_BYTES, sizeof(PeerBlob));
        bool all = true;
        uint64_t token = 1469598103934665603ull;
        for (int r = 0; r < world_size; ++r) {
            const PeerBlob& b = bl[(size_t)r];
            if (!(b.flags & 1u)) all = false;
            uint64_t rows = 0, budget = ~0ull; int here = 0;
            for (int q = 0; q < world_size; ++q)
                if (bl[(size_t)q].dev_id == b.dev_id) { rows += bl[(size_t)q].sh_rows; if (bl[(size_t)q].budget_shared < budget) budget = bl[(size_t)q].budget_shared; ++here; }
            if (here > 1 && rows > budget) all = false;
            for (uint64_t v : {(uint64_t)(uint32_t)b.pid, b.ptr}) { token ^= v; token *= 1099511628211ull; }
        }
        c->group_persist = all;
        c->group_token = token | 1ull;
        c->coop_allowed = true;
        for (int r = 0; r < world_size; ++r)
            if (r != rank && bl[(size_t)r].pid == bl[(size_t)rank].pid && bl[(size_t)r].dev_id == bl[(size_t)rank].dev_id) c->coop_allowed = false;
        c->coop_validated = false; c->persist_refused = false;
        c->peer_share = 0;
        for (int r = 0; r < world_size; ++r) if (bl[(size_t)r].dev_id == bl[(size_t)rank].dev_id) c->peer_share += 1;
    }
    c->world_size = world_size; c->rank = rank; c->multi = true;
    return RSRL_HIP_OK;
}

// ---- single-process group: every rank is a ctx of THIS process (SURVEY 8b last row; the reference's owner graph is single-threaded,
// rsrl/src/core.rs:13-15, so a Rust host cannot run one blocking ncclCommInitRank per ctx).  One call attaches an exchange to all
// of them, rank = index:
//   RSRL_EXCHANGE_PEER  export + connect of every ctx (same-process pointers; peer access enabled between the devices)
//   RSRL_EXCHANGE_RCCL  ncclCommInitAll over the ctxs' devices (distinct devices, RCCL's rule), then one grouped warm-up
//                       all-reduce so that connection set-up, which needs every rank, does not happen inside the first train()
// Afterwards a single host thread drives the ranks by calling rsrl_hip_train on each ctx in turn: the calls only enqueue.
int rsrl_hip_group_create(rsrl_hip_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1 || n > 64) return fail(RSRL_HIP_EINVAL, "bad group arguments");
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        if (!c) return fail(RSRL_HIP_EINVAL, "null ctx in the group");
        for (int j = 0; j < i; ++j) if (ctxs[j] == c) return fail(RSRL_HIP_EINVAL, "ctx %d appears twice in the group", i);
        FLUSH(c);
        if (c->cfg.weight_mode != RSRL_W_SHARED) return fail(RSRL_HIP_ESTATE, "per-env weights need no exchange: shard by env_offset instead");
        if (c->multi || c->comm || c->peer_recv) return fail(RSRL_HIP_ESTATE, "ctx %d already has an exchange attached", i);
        if (c->cfg.exchange != ctxs[0]->cfg.exchange || c->dw_elems != ctxs[0]->dw_elems || c->cfg.basis != ctxs[0]->cfg.basis)
            return fail(RSRL_HIP_EINVAL, "the ctxs of a group must share the approximator's shape and the exchange kind");
    }
    if (ctxs[0]->cfg.exchange == RSRL_EXCHANGE_AUTO) {
        // PEER whenever every device of the group reaches every other one's memory (one hop, exact integer sums, the persistent kernel);
        // RCCL, the any-topology fallback, otherwise -- or when ranks share a device but RCCL was not asked for explicitly (it needs one
        // device per rank)
        bool peer = true;
        for (int i = 0; i < n && peer; ++i)
            for (int j = 0; j < n && peer; ++j) {
                const int a = ctxs[i]->cfg.device, b = ctxs[j]->cfg.device;
                if (a == b) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess) { (void)hipGetLastError(); can = 0; }
                if (!can) peer = false;
            }
        for (int i = 0; i < n; ++i) ctxs[i]->cfg.exchange = peer ? RSRL_EXCHANGE_PEER : RSRL_EXCHANGE_RCCL;
    }
    if (ctxs[0]->cfg.exchange == RSRL_EXCHANGE_PEER) {
        std::vector<uint8_t> handles((size_t)n * RSRL_HIP_PEER_HANDLE_BYTES);
        for (int i = 0; i < n; ++i) TRY(rsrl_hip_peer_export(ctxs[i], n, handles.data() + (size_t)i * RSRL_HIP_PEER_HANDLE_BYTES));
        for (int i = 0; i < n; ++i) TRY(rsrl_hip_peer_connect(ctxs[i], handles.data(), n, i));
        return RSRL_HIP_OK;
    }
    std::vector<int> devs((size_t)n);
    for (int i = 0; i < n; ++i) {
        devs[(size_t)i] = ctxs[i]->cfg.device;
        for (int j = 0; j < i; ++j)
            // (a collectives library that admits ranks sharing a device says so by exporting `rccl_stub_allows_shared_device` -- the test double of
            // tests/stubs/rccl_stub.cpp, LD_PRELOADed, which exercises this path on a one-GPU box; real RCCL has no such symbol and the check stands)
            if (devs[(size_t)j] == devs[(size_t)i] && !dlsym(RTLD_DEFAULT, "rccl_stub_allows_shared_device"))
                return fail(RSRL_HIP_EINVAL, "RCCL needs one device per rank: ctxs %d and %d share device %d (use RSRL_EXCHANGE_PEER)", j, i, devs[(size_t)i]);
    }
    std::vector<ncclComm_t> comms((size_t)n, nullptr);
    NCCL_TRY(ncclCommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; ++i) { ctxs[i]->comm = comms[(size_t)i]; ctxs[i]->world_size = n; ctxs[i]->rank = i; ctxs[i]->multi = true; ctxs[i]->st_rccl_group = n > 1; }
    // warm-up: dW is zero between operations, so the grouped all-reduce leaves it zero
    NCCL_TRY(ncclGroupStart());
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        HIP_TRY(hipSetDevice(c->cfg.device));
        NCCL_TRY(ncclAllReduce(c->dW, c->dW, c->dw_elems, ncclFloat, ncclSum, c->comm, c->stream));
    }
    NCCL_TRY(ncclGroupEnd());
    for (int i = 0; i < n; ++i) { HIP_TRY(hipSetDevice(ctxs[i]->cfg.device)); HIP_TRY(hipStreamSynchronize(ctxs[i]->stream)); }
    return RSRL_HIP_OK;
}

// One thread stepping every rank of a group it created with rsrl_hip_group_create.  What makes this an entry point of its own:
//  * RCCL: a thread that drives several communicators must issue each collective for ALL of them inside one ncclGroupStart / End;
//    un-grouped, an all-reduce of rank 0 may wait for a rank the same thread has not reached yet, and a train call enqueues hundreds
//    of them.  So the batch-steps advance in lock-step here -- every rank's step kernels, then every rank's all-reduce in ONE group,
//    then what follows the exchange -- and rsrl_hip_train / rsrl_hip_handle refuse such a ctx (RSRL_HIP_ESTATE).
//  * PEER: a rank's exchange kernels wait (bounded) for its peers' kernels, which this same thread has yet to enqueue: the ranks
//    are therefore fed in turns of at most 32 batch-steps, so that no rank's launch queue can fill up in front of a peer that has
//    nothing enqueued (the persistent kernel is one launch per rank and call: no turns needed).
// Results are those of rsrl_hip_train on every rank from a thread of its own, bit for bit.
int rsrl_hip_group_train(rsrl_hip_ctx* const* ctxs, int n, int64_t n_steps) {
    if (!ctxs || n < 1 || n > 64) return fail(RSRL_HIP_EINVAL, "bad group arguments");
    if (n_steps < 0) return fail(RSRL_HIP_EINVAL, "n_steps < 0");
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        if (!c) return fail(RSRL_HIP_EINVAL, "null ctx in the group");
        if (!c->multi || c->world_size != n || c->rank != i || c->cfg.exchange != ctxs[0]->cfg.exchange ||
            (c->cfg.exchange == RSRL_EXCHANGE_PEER && c->group_token != ctxs[0]->group_token))
            return fail(RSRL_HIP_ESTATE, "ctxs[0..%d) must be exactly the ranks of one group made by rsrl_hip_group_create, in rank order", n);
        FLUSH(c);
    }
    if (n_steps == 0) return RSRL_HIP_OK;
    if (ctxs[0]->cfg.exchange == RSRL_EXCHANGE_PEER) {
        const int64_t turn = persist_ok(ctxs[0]) ? n_steps : kStepsPerGraph;
        for (int64_t done = 0; done < n_steps; done += turn)
            for (int i = 0; i < n; ++i) TRY(train_now(ctxs[i], n_steps - done < turn ? n_steps - done : turn, nullptr));
        return RSRL_HIP_OK;
    }
    // RCCL: lock-step, the all-reduces of a batch-step grouped
    std::vector<Common> ks((size_t)n);
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        HIP_TRY(hipSetDevice(c->cfg.device));
        if (c->sh_tab && c->t != c->sh_tab_t) HIP_TRY(hipMemsetAsync(c->sh_tab, 0, sizeof(long long) * 3 * kTabRep * c->dw_elems, c->stream));
    }
    for (int64_t j = 0; j < n_steps; ++j) {
        for (int i = 0; i < n; ++i) {
            rsrl_hip_ctx* c = ctxs[i];
            HIP_TRY(hipSetDevice(c->cfg.device));
            ks[(size_t)i] = make_common(c);
            TRY(enqueue_shared_step(c, ks[(size_t)i], make_geom(c), nullptr, j == 0 ? 0 : 1, c->t, nullptr, 1));
        }
        NCCL_TRY(ncclGroupStart());
        for (int i = 0; i < n; ++i) {
            rsrl_hip_ctx* c = ctxs[i];
            HIP_TRY(hipSetDevice(c->cfg.device));
            if (c->sh_tab) TRY(exchange_table(c, c->t));                  // dense basis: the batch-step's fixed-point table
            else NCCL_TRY(ncclAllReduce(c->dW, c->dW, c->dw_elems, ncclFloat, ncclSum, c->comm, c->stream));      // tile coding: the float delta
        }
        NCCL_TRY(ncclGroupEnd());
        for (int i = 0; i < n; ++i) {
            rsrl_hip_ctx* c = ctxs[i];
            HIP_TRY(hipSetDevice(c->cfg.device));
            TRY(enqueue_shared_step(c, ks[(size_t)i], make_geom(c), nullptr, j == 0 ? 0 : 1, c->t, nullptr, 2));
            c->t += 1;
            c->kernel_name = shared_kernel_name(c);
        }
    }
    for (int i = 0; i < n; ++i) {
        rsrl_hip_ctx* c = ctxs[i];
        HIP_TRY(hipSetDevice(c->cfg.device));
        TRY(enqueue_shared_c(c, make_common(c), make_geom(c), c->t - 1));
        c->sh_tab_t = c->t;
    }
    return RSRL_HIP_OK;
}
RSRL_API_END
