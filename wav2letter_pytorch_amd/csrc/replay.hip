// Recorded launch lists (round 6): the step engine records the entry points one training step calls -- which function, its
// argument values, on which stream, with which event records / waits between the streams -- ONCE, after the tuned warm-up,
// and from then on replays each phase of the step (forward, backward, optimizer, held-back weight gradients) through ONE
// call of w2l_replay: a C loop over the very same extern "C" entry points.  No hipGraph: every launch still goes to the
// stream it was recorded on, so the side streams, the cross-step overlap of the weight updates and the held-back weight
// gradients work exactly as in the eager step; what disappears is the ~200-500 Python -> ctypes transitions per step
// (4.4-5.0 ms of host time per Wav2Letter step, 12.3 ms per Jasper 10x5 step: DESIGN Appendix C).
//
// Replaces nothing in the reference (its host loop is PyTorch's dispatcher); the recordable primitives below stand in for the
// torch.cuda.Event / Stream.wait_event / tensor.zero_() / torch._foreach_* calls the eager engine made between launches.
#include "common.h"
#include "../../include/w2l_hip.h"
#include <string.h>
#include <mutex>
#include <type_traits>
#include <utility>
#include <vector>

// ---- events and stream order as entry points (recordable like any launch) -----------------------------------------------
extern "C" int w2l_event_create(void** ev) {
    W2L_CHECK_ARG(ev != nullptr, "event_create: null pointer");
    hipEvent_t e;
    W2L_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *ev = (void*)e;
    return 0;
}
extern "C" int w2l_event_destroy(void* ev) {
    if (ev != nullptr) W2L_CHECK_HIP(hipEventDestroy((hipEvent_t)ev));
    return 0;
}
extern "C" int w2l_event_record(void* ev, void* stream) {
    W2L_CHECK_ARG(ev != nullptr, "event_record: null event");
    W2L_CHECK_HIP(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream));
    return 0;
}
extern "C" int w2l_stream_wait_event(void* stream, void* ev) {
    W2L_CHECK_ARG(ev != nullptr, "stream_wait_event: null event");
    W2L_CHECK_HIP(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return 0;
}
// 0 = the event has fired, 1 = not yet, anything else = a HIP error (w2l_last_error)
extern "C" int w2l_event_query(void* ev) {
    W2L_CHECK_ARG(ev != nullptr, "event_query: null event");
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) { (void)hipGetLastError(); return 1; }
    w2l_set_error("hipEventQuery failed: %s", hipGetErrorString(e));
    return (int)e;
}
extern "C" int w2l_event_synchronize(void* ev) {
    W2L_CHECK_ARG(ev != nullptr, "event_synchronize: null event");
    W2L_CHECK_HIP(hipEventSynchronize((hipEvent_t)ev));
    return 0;
}
// `waiter` waits for everything enqueued on `signaler` so far.  The event comes from a small per-thread ring: a wait
// captures the event's state at the call, so the event may be recorded again at once.
extern "C" int w2l_stream_wait_stream(void* waiter, void* signaler) {
    if (waiter == signaler) return 0;
    constexpr int kRing = 64;
    static thread_local hipEvent_t ring[kRing];
    static thread_local int created = 0, next = 0;
    if (created < kRing && next == created) {
        W2L_CHECK_HIP(hipEventCreateWithFlags(&ring[created], hipEventDisableTiming));
        ++created;
    }
    hipEvent_t e = ring[next];
    next = (next + 1) % kRing;
    W2L_CHECK_HIP(hipEventRecord(e, (hipStream_t)signaler));
    W2L_CHECK_HIP(hipStreamWaitEvent((hipStream_t)waiter, e, 0));
    return 0;
}

// ---- the few elementwise odds and ends the eager engine left to torch ops ------------------------------------------------
extern "C" int w2l_fill_zero(void* p, int64_t bytes, void* stream) {
    W2L_CHECK_ARG(bytes >= 0 && (p != nullptr || bytes == 0), "fill_zero: null pointer");
    if (bytes > 0) W2L_CHECK_HIP(hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream));
    return 0;
}

namespace {
__global__ void pad_vec_kernel(const float* src, int n, float* dst, int cp, float fill) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cp) dst[i] = i < n ? src[i] : fill;
}
__global__ void counter_add_kernel(long long* c, long long delta) {
    if (threadIdx.x == 0 && blockIdx.x == 0) c[0] += delta;
}
__global__ void add_i64_multi_kernel(long long* const* table, int n, long long delta) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) table[i][0] += delta;
}
// torch.optim.SGD's update of one small parameter (bias, BatchNorm gamma / beta) per block: g' = g + wd p;
// m = mu m + g' (dampening 0);  p -= lr (nesterov ? g' + mu m : m)
// grid = (items, chunks of SGD_SMALL_CHUNK elements): most items are one chunk (a few hundred channels); the classifier weight
// (29 x 1024) is fifteen -- one block walking it alone took 96 us on the caller's stream at every step boundary
constexpr int SGD_SMALL_CHUNK = 2048;
__global__ __launch_bounds__(256) void sgd_small_multi_kernel(const w2l_sgd_small_t* items, float lr, float mu, float wd, int nesterov) {
    const w2l_sgd_small_t it = items[blockIdx.x];
    const int lo = blockIdx.y * SGD_SMALL_CHUNK, hi = min(it.n, lo + SGD_SMALL_CHUNK);
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        float pv = it.p[i];
        float gv = it.g[i];
        if (wd != 0.f) gv += wd * pv;
        float step = gv;
        if (it.m != nullptr) {
            const float mv = mu * it.m[i] + gv;
            it.m[i] = mv;
            step = nesterov ? gv + mu * mv : mv;
        }
        it.p[i] = pv - lr * step;
    }
}
}  // namespace

extern "C" int w2l_pad_vec_f32(const float* src, int n, float* dst, int cp, float fill, void* stream) {
    W2L_CHECK_ARG(src != nullptr && dst != nullptr && n >= 0 && cp >= n, "pad_vec_f32: bad arguments");
    if (cp == 0) return 0;
    hipLaunchKernelGGL(pad_vec_kernel, dim3((cp + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, n, dst, cp, fill);
    W2L_CHECK_LAUNCH();
    return 0;
}
extern "C" int w2l_counter_add(void* counter_i64, int64_t delta, void* stream) {
    W2L_CHECK_ARG(counter_i64 != nullptr, "counter_add: null pointer");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)counter_i64, (long long)delta);
    W2L_CHECK_LAUNCH();
    return 0;
}
extern "C" int w2l_add_i64_multi(void* table_dev, int n, int64_t delta, void* stream) {
    W2L_CHECK_ARG(table_dev != nullptr || n == 0, "add_i64_multi: null table");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(add_i64_multi_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, (long long* const*)table_dev, n,
                       (long long)delta);
    W2L_CHECK_LAUNCH();
    return 0;
}
extern "C" int w2l_sgd_small_multi(const w2l_sgd_small_t* items_dev, int nitems, int max_n, float lr, float momentum,
                                   float weight_decay, int nesterov, void* stream) {
    W2L_CHECK_ARG(items_dev != nullptr || nitems == 0, "sgd_small_multi: null table");
    W2L_CHECK_ARG(max_n >= 0 && max_n <= (1 << 26), "sgd_small_multi: max_n (the largest item's element count) out of range");
    if (nitems <= 0 || max_n == 0) return 0;
    const int chunks = (max_n + SGD_SMALL_CHUNK - 1) / SGD_SMALL_CHUNK;
    hipLaunchKernelGGL(sgd_small_multi_kernel, dim3(nitems, chunks), dim3(256), 0, (hipStream_t)stream, items_dev, lr, momentum,
                       weight_decay, nesterov);
    W2L_CHECK_LAUNCH();
    return 0;
}

// ---- the replay loop ------------------------------------------------------------------------------------------------------
namespace {
template <class T>
inline T slot_get(const w2l_slot_t& s) {
    if constexpr (std::is_pointer_v<T>) return (T)s.p;
    else if constexpr (std::is_floating_point_v<T>) return (T)s.d;
    else return (T)s.i;
}
template <class F>
struct Sig;
template <class R, class... A>
struct Sig<R (*)(A...)> {
    static constexpr int arity = (int)sizeof...(A);
    template <R (*Fn)(A...), size_t... I>
    static int call(const w2l_slot_t* a, std::index_sequence<I...>) {
        if constexpr (std::is_void_v<R>) {
            Fn(slot_get<A>(a[I])...);
            return 0;
        } else {
            return (int)Fn(slot_get<A>(a[I])...);
        }
    }
};
template <auto Fn>
int thunk(const w2l_slot_t* a) {
    using S = Sig<decltype(Fn)>;
    return S::template call<Fn>(a, std::make_index_sequence<(size_t)S::arity>{});
}
struct Entry {
    const char* name;
    int (*fn)(const w2l_slot_t*);
    int arity;
};
#define W2L_E(f) {#f, &thunk<&f>, Sig<decltype(&f)>::arity}
// every entry point a training step can call between two host decisions (host-only queries are not here: their results
// are part of the recorded control flow)
const Entry kEntries[] = {
    W2L_E(w2l_event_record), W2L_E(w2l_stream_wait_event), W2L_E(w2l_stream_wait_stream), W2L_E(w2l_fill_zero),
    W2L_E(w2l_pad_vec_f32), W2L_E(w2l_counter_add), W2L_E(w2l_add_i64_multi), W2L_E(w2l_sgd_small_multi),
    W2L_E(w2l_conv_stats_mode), W2L_E(w2l_wgrad_deterministic),
    W2L_E(w2l_pack_weights), W2L_E(w2l_sgd_pack), W2L_E(w2l_novograd_pack), W2L_E(w2l_nct_to_ntc), W2L_E(w2l_pad_cast),
    W2L_E(w2l_conv1d_igemm), W2L_E(w2l_conv1d_igemm_ws), W2L_E(w2l_conv1d_igemm_fp8), W2L_E(w2l_conv1d_dgrad_bnreduce_ws),
    W2L_E(w2l_conv1d_wgrad), W2L_E(w2l_conv1d_wgrad_ws), W2L_E(w2l_conv1d_wgrad_group), W2L_E(w2l_conv1d_wgrad_fp8),
    W2L_E(w2l_dwconv_fwd), W2L_E(w2l_dwconv_dgrad), W2L_E(w2l_dwconv_wgrad),
    W2L_E(w2l_bn_finalize), W2L_E(w2l_bn_act_fwd), W2L_E(w2l_bn_act_fwd_q), W2L_E(w2l_bn_act_fwd_fin),
    W2L_E(w2l_bn_act_bwd_reduce), W2L_E(w2l_bn_act_bwd_reduce_slots), W2L_E(w2l_bn_bwd_finalize), W2L_E(w2l_bn_act_bwd_apply),
    W2L_E(w2l_bn_act_bwd_apply_amax), W2L_E(w2l_bn_act_bwd_apply_fin), W2L_E(w2l_bn_act_bwd_apply_slots),
    W2L_E(w2l_quantize_e4m3), W2L_E(w2l_quantize_e4m3_dyn),
    W2L_E(w2l_log_softmax_fwd), W2L_E(w2l_log_softmax_bwd), W2L_E(w2l_ctc_loss), W2L_E(w2l_argmax),
    W2L_E(w2l_rccl_all_reduce),
};
constexpr int kNumEntries = (int)(sizeof(kEntries) / sizeof(kEntries[0]));
}  // namespace

extern "C" int w2l_replay_op(const char* name) {
    if (name == nullptr) return -1;
    for (int i = 0; i < kNumEntries; ++i)
        if (strcmp(kEntries[i].name, name) == 0) return i;
    return -1;
}
extern "C" int w2l_replay_arity(int op) { return op >= 0 && op < kNumEntries ? kEntries[op].arity : -1; }

extern "C" int w2l_replay(const w2l_call_t* calls, int n, int* failed_at) {
    W2L_CHECK_ARG(calls != nullptr || n == 0, "replay: null list");
    for (int i = 0; i < n; ++i) {
        const w2l_call_t& c = calls[i];
        if (c.op < 0 || c.op >= kNumEntries || c.nargs != kEntries[c.op].arity) {
            if (failed_at) *failed_at = i;
            w2l_set_error("replay: record %d names entry point %d with %d arguments", i, c.op, c.nargs);
            return 1;
        }
        const int rc = kEntries[c.op].fn(c.a);
        if (rc != 0) {                       // (the entry point has set the error string)
            if (failed_at) *failed_at = i;
            return rc;
        }
    }
    return 0;
}
