// One-shot all-reduce(SUM) of the trainable-gradient buffer among the GPUs of one node (SURVEY.md sections 5, 8(b), 8(e):
// `ynet_comm_init` / `ynet_allreduce_sum`).
//
// The data-parallel step exchanges ONE small flat buffer (32 KB for mosa_1, 131 KB for mosa_4, 6.6 MB with every weight
// trainable): latency-bound.  A ring all-reduce takes 2 (N - 1) dependent hops; here every rank publishes its buffer in
// a mailbox that the other ranks map through HIP IPC (peer access over xGMI) and each rank reads the N - 1 peers
// directly -- one hop -- summing in RANK ORDER, so every rank obtains bit-identical sums (the replicated Adam updates
// stay in lock-step without a broadcast) and the result does not depend on arrival order.
//
//   mailbox (fine-grained device memory, exported with hipIpcGetMemHandle):
//       flag[2]            epoch number of the data in slot 0 / 1, written with a system-scope release
//       slot[2][capacity]  double buffer: call number e uses slot e & 1
//   call e (epochs count from 1 in a DEVICE-resident counter, equal on every rank because the calls are collective; every
//   workgroup reads it at entry, the last workgroup to publish advances it):
//       1. every workgroup copies its part of buf into slot[e & 1]; the last one to finish publishes flag[e & 1] = e
//       2. every workgroup waits until each peer's flag[e & 1] == e (system-scope acquire), then adds the peers' parts to
//          its own in rank order and writes buf
//   (the 32-bit call number wraps 0xFFFFFFFF -> 2, which keeps the slots alternating)
//   Slot reuse is safe with two slots: a rank enters call e only after call e - 1 returned on its stream, call e - 1
//   needed every peer's flag e - 1, and a peer publishes flag e - 1 only after ITS call e - 2 finished reading.
// A peer that never arrives (a crashed rank) would spin forever: the wait gives up after ~20 s of s_memrealtime (every
// workgroup on its own), records the failure in the mailbox and POISONS its part of buf with NaN; the workgroup that
// FINISHES LAST (a ticket) poisons the loss slot when any workgroup failed -- after every other write to it --, so the
// step cannot silently continue on un-reduced gradients; ynet_comm_status reports it (dist.DataParallel.check raises).
// The launch has no per-call argument (the epoch lives on the device), so it can be captured INTO a hipGraph:
// utils/step_graph.py records it between the backward pass and the optimizer step -- one graph per step.
#include "ynet_common.h"
#include <stdlib.h>
#include <string.h>

#define YNET_COMM_MAX_RANKS 16
#define YNET_COMM_HEADER_BYTES 256      // flags, error word; the slots start 256-byte aligned

struct YnetComm {
    int rank, world;
    long long capacity;                 // floats per slot
    unsigned long long epoch;
    unsigned char* local;               // this rank's mailbox (device)
    unsigned char* peer[YNET_COMM_MAX_RANKS];
    unsigned* done_counter;             // device: [0] workgroups that finished copying, [1] workgroups that finished the call,
                                        //   [2] a workgroup of this call timed out (all three reset by the kernel), [3] calls so far
    int connected;
};

struct AllreduceArgs {
    float* buf;
    long long n;
    unsigned char* mbox[YNET_COMM_MAX_RANKS];
    long long capacity;
    unsigned* done_counter;
    int rank, world;
};

__global__ __launch_bounds__(256) void allreduce_oneshot_kernel(const AllreduceArgs a) {
    // the call number: read by every workgroup before it publishes; advanced by the last workgroup to publish (below), i.e.
    // after all of them have read it (0 is the flags' initial value and is skipped)
    unsigned epoch = __hip_atomic_load(a.done_counter + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    if (epoch == 0u) epoch = 2u;      // (wrap after 2^32 - 1 calls: 0 is the flags' initial value; 2, not 1 -- call 0xFFFFFFFF used slot 1, and two consecutive calls must not share a slot)
    epoch = __builtin_amdgcn_readfirstlane(epoch);
    const int slot = (int)(epoch & 1u);
    unsigned* my_flags = reinterpret_cast<unsigned*>(a.mbox[a.rank]);
    float* my_slot = reinterpret_cast<float*>(a.mbox[a.rank] + YNET_COMM_HEADER_BYTES) + (long long)slot * a.capacity;
    const long long per = ((a.n + gridDim.x - 1) / gridDim.x + 3) & ~3ll;      // this workgroup's part (multiple of 4 floats)
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < a.n ? lo + per : a.n;
    // ---- 1. publish
    for (long long i = lo + threadIdx.x; i < hi; i += 256) my_slot[i] = a.buf[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned arrived = atomicAdd(a.done_counter, 1u) + 1u;
        if (arrived == gridDim.x) {
            *a.done_counter = 0u;
            __hip_atomic_store(a.done_counter + 3, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(my_flags + slot, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    // ---- 2. wait for every peer's epoch, then reduce in rank order
    __shared__ int failed;
    if (threadIdx.x == 0) {
        failed = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();       // 100 MHz
        for (int r = 0; r < a.world && !failed; ++r) {
            if (r == a.rank) continue;
            const unsigned* f = reinterpret_cast<const unsigned*>(a.mbox[r]) + slot;
            while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
                __builtin_amdgcn_s_sleep(8);
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {      // ~20 s
                    failed = 1;
                    my_flags[2] = 1u;       // error word (read by ynet_comm_status)
                    break;
                }
            }
        }
    }
    __syncthreads();
    if (failed) {
        // A peer never arrived: this workgroup's part of buf cannot be reduced.  Leaving it un-reduced would let the ranks
        // apply DIFFERENT gradients silently (the replicated Adam states diverge), so the part is poisoned with NaN; the loss
        // that rides in the last slot is poisoned by the call's last workgroup (below): the step fails loudly on this rank
        // (NaN loss, NaN weights), and ynet_comm_status / DataParallel.check() report the time-out at the next
        // synchronisation point.
        const float poison = __builtin_nanf("");
        for (long long i = lo + threadIdx.x; i < hi; i += 256) a.buf[i] = poison;
    } else {
        __threadfence_system();
        for (long long i = lo + threadIdx.x; i < hi; i += 256) {
            float s = 0.f;
            for (int r = 0; r < a.world; ++r) {     // rank order on every rank: identical sums everywhere
                const float* ps = reinterpret_cast<const float*>(a.mbox[r] + YNET_COMM_HEADER_BYTES) + (long long)slot * a.capacity;
                s += r == a.rank ? a.buf[i] : __builtin_nontemporal_load(ps + i);
            }
            a.buf[i] = s;
        }
    }
    // ---- 3. the loss slot is decided by ONE workgroup, the last to finish: a failed workgroup's NaN and the owning
    // workgroup's sum can then not race
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        if (failed) atomicOr(a.done_counter + 2, 1u);
        const unsigned finished = atomicAdd(a.done_counter + 1, 1u) + 1u;
        if (finished == gridDim.x) {
            if (atomicExch(a.done_counter + 2, 0u) != 0u) a.buf[a.n - 1] = __builtin_nanf("");
            a.done_counter[1] = 0u;
        }
    }
}

extern "C" {

long long ynet_comm_handle_bytes(void) { return (long long)sizeof(hipIpcMemHandle_t); }

int ynet_comm_create(int rank, int world, long long max_floats, void** comm_out) {
    YNET_REQUIRE(comm_out != nullptr, "comm_create: null pointer");
    YNET_REQUIRE(world >= 1 && world <= YNET_COMM_MAX_RANKS && rank >= 0 && rank < world, "comm_create: rank %d of %d (at most %d ranks)", rank, world, YNET_COMM_MAX_RANKS);
    YNET_REQUIRE(max_floats > 0 && max_floats < (1ll << 31), "comm_create: capacity %lld floats", max_floats);
    YnetComm* c = (YnetComm*)calloc(1, sizeof(YnetComm));
    YNET_REQUIRE(c != nullptr, "comm_create: out of host memory");
    c->rank = rank;
    c->world = world;
    c->capacity = (max_floats + 63) & ~63ll;
    const size_t bytes = YNET_COMM_HEADER_BYTES + 2 * (size_t)c->capacity * sizeof(float);
    // fine-grained: peers must observe the data while the kernels of both sides are still running
    hipError_t e = hipExtMallocWithFlags((void**)&c->local, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        free(c);
        ynet_set_error("comm_create: hipExtMallocWithFlags(%zu bytes, fine-grained) failed: %s", bytes, hipGetErrorString(e));
        return 2;
    }
    e = hipMalloc((void**)&c->done_counter, 64);
    if (e == hipSuccess) e = hipMemset(c->local, 0, YNET_COMM_HEADER_BYTES);
    if (e == hipSuccess) e = hipMemset(c->done_counter, 0, 64);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        (void)hipFree(c->local);
        if (c->done_counter) (void)hipFree(c->done_counter);
        free(c);
        ynet_set_error("comm_create: initialisation failed: %s", hipGetErrorString(e));
        return 2;
    }
    c->peer[rank] = c->local;
    *comm_out = c;
    return 0;
}

int ynet_comm_export(void* comm, void* handle_out) {
    YnetComm* c = (YnetComm*)comm;
    YNET_REQUIRE(c && handle_out, "comm_export: null pointer");
    hipIpcMemHandle_t h;
    const hipError_t e = hipIpcGetMemHandle(&h, c->local);
    if (e != hipSuccess) {
        ynet_set_error("comm_export: hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set)", hipGetErrorString(e));
        return 2;
    }
    memcpy(handle_out, &h, sizeof(h));
    return 0;
}

int ynet_comm_connect(void* comm, const void* handles) {
    YnetComm* c = (YnetComm*)comm;
    YNET_REQUIRE(c && handles, "comm_connect: null pointer");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)r * sizeof(h), sizeof(h));
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            ynet_set_error("comm_connect: hipIpcOpenMemHandle(rank %d) failed: %s", r, hipGetErrorString(e));
            return 2;
        }
        c->peer[r] = (unsigned char*)p;
    }
    c->connected = 1;
    return 0;
}

int ynet_allreduce_sum(void* comm, float* buf, long long n, void* stream) {
    YnetComm* c = (YnetComm*)comm;
    YNET_REQUIRE(c && buf, "allreduce_sum: null pointer");
    YNET_REQUIRE(c->world == 1 || c->connected, "allreduce_sum: ynet_comm_connect has not been called");
    YNET_REQUIRE(n > 0 && n <= c->capacity, "allreduce_sum: %lld floats exceed the mailbox capacity %lld", n, c->capacity);
    // (a world of one rank still launches: publish, no peer to wait for, sum = the input bit for bit -- the forced
    // single-rank runs of dist.DataParallel(force=True) record this launch into the step's graph like N ranks do)
    AllreduceArgs a{};
    a.buf = buf;
    a.n = n;
    for (int r = 0; r < c->world; ++r) a.mbox[r] = c->peer[r];
    a.capacity = c->capacity;
    a.done_counter = c->done_counter;
    ++c->epoch;                                               // (host-side count of the calls issued; the kernel keeps its own)
    a.rank = c->rank;
    a.world = c->world;
    int blocks = (int)((n + 4095) / 4096);                    // 16 floats per thread; a few workgroups for the large buffers
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    return ynet_check_launch("allreduce_sum");
}

/* 0 = healthy, 1 = a wait for a peer timed out in some earlier call (synchronises the device) */
int ynet_comm_status(void* comm) {
    YnetComm* c = (YnetComm*)comm;
    if (!c) return 1;
    unsigned word = 0;
    (void)hipDeviceSynchronize();
    if (hipMemcpy(&word, c->local + 2 * sizeof(unsigned), sizeof(word), hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return word ? 1 : 0;
}

int ynet_comm_destroy(void* comm) {
    YnetComm* c = (YnetComm*)comm;
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r)
        if (r != c->rank && c->peer[r]) (void)hipIpcCloseMemHandle(c->peer[r]);
    (void)hipFree(c->local);
    (void)hipFree(c->done_counter);
    free(c);
    return 0;
}

}  // extern "C"
