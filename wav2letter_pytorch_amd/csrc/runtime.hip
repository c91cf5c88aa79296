// Error reporting, ABI version and host-side helpers of libw2l_hip.so.
#include "common.h"
#include <mutex>
#include <string.h>
#include <unordered_set>
#include <vector>
#include <algorithm>

static thread_local char g_err[512] = "";

void w2l_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

hipError_t w2l_allow_big_lds(const void* kernel) {
    static std::mutex mu;
    static std::unordered_set<const void*> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(kernel)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert(kernel);
    return e;
}

extern "C" const char* w2l_last_error(void) { return g_err; }
extern "C" int w2l_abi_version(void) { return 2; }        // 2: w2l_bnact_t gained q_clipped

// ---- tuning cache persistence (the measured block-shape / split-K choices of w2l_conv1d_*_tune) ----
void w2l_igemm_tune_dump(FILE* f);
bool w2l_igemm_tune_put(const int* v);
void w2l_wgrad_tune_dump(FILE* f);
bool w2l_wgrad_tune_put(const int* v);
void w2l_wgrad_fp8_tune_dump(FILE* f);
void w2l_igemm_fp8_tune_dump(FILE* f);
bool w2l_igemm_fp8_tune_put(const int* v);
bool w2l_wgrad_fp8_tune_put(const int* v);
static const char kTuneHeader[] = "w2l-tune v2 gfx950";      // v2: 26 implicit-GEMM block shapes (configuration indices moved)

extern "C" int w2l_tune_save(const char* path) {
    W2L_CHECK_ARG(path != nullptr, "tune_save: null path");
    FILE* f = fopen(path, "w");
    W2L_CHECK_ARG(f != nullptr, "tune_save: cannot open %s", path);
    fprintf(f, "%s\n", kTuneHeader);
    w2l_igemm_tune_dump(f);
    w2l_wgrad_tune_dump(f);
    w2l_wgrad_fp8_tune_dump(f);
    w2l_igemm_fp8_tune_dump(f);
    const bool ok = fclose(f) == 0;
    W2L_CHECK_ARG(ok, "tune_save: write to %s failed", path);
    return 0;
}

// Returns the number of entries taken (>= 0), or -1 on error.  Unknown / infeasible lines are skipped, so a cache
// written by another build can never select an invalid launch.
extern "C" int w2l_tune_load(const char* path) {
    if (path == nullptr) { w2l_set_error("tune_load: null path"); return -1; }
    FILE* f = fopen(path, "r");
    if (f == nullptr) { w2l_set_error("tune_load: cannot open %s", path); return -1; }
    char line[256];
    int taken = 0;
    if (fgets(line, sizeof(line), f) == nullptr || strncmp(line, kTuneHeader, sizeof(kTuneHeader) - 1) != 0) {
        fclose(f);
        w2l_set_error("tune_load: %s is not a %s file", path, kTuneHeader);
        return -1;
    }
    while (fgets(line, sizeof(line), f) != nullptr) {
        int v[9];
        if (sscanf(line, "igemm %d %d %d %d %d %d %d %d %d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8) == 9)
            taken += w2l_igemm_tune_put(v) ? 1 : 0;
        else if (sscanf(line, "wgrad %d %d %d %d %d %d %d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6) == 7)
            taken += w2l_wgrad_tune_put(v) ? 1 : 0;
        else if (sscanf(line, "igemmf8 %d %d %d %d %d %d %d %d %d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7, v + 8) == 9)
            taken += w2l_igemm_fp8_tune_put(v) ? 1 : 0;
        else if (sscanf(line, "wgradf8 %d %d %d %d %d %d %d", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6) == 7)
            taken += w2l_wgrad_fp8_tune_put(v) ? 1 : 0;
    }
    fclose(f);
    return taken;
}

// Edit distance over int32 symbols (stands in for python-Levenshtein's distance(),
// decoder.py:49,60).  Host code: the strings live on the host.
extern "C" int w2l_levenshtein_host(const int32_t* a, int na, const int32_t* b, int nb) {
    if (na < 0 || nb < 0) return -1;
    if (na < nb) { std::swap(a, b); std::swap(na, nb); }
    std::vector<int> prev(nb + 1), cur(nb + 1);
    for (int j = 0; j <= nb; ++j) prev[j] = j;
    for (int i = 1; i <= na; ++i) {
        cur[0] = i;
        for (int j = 1; j <= nb; ++j)
            cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + (a[i - 1] != b[j - 1]));
        std::swap(prev, cur);
    }
    return prev[nb];
}

// ---- greedy decode + CER / WER totals of one batch, host side (base_asr_models.py:53-69 with decoder.py:31-66,104-119) ----
// What ConvCTCASR.add_string_metrics does per step, as ONE call that never takes the interpreter lock: the step's argmax
// indices (already on the host: one asynchronous copy per step) are collapsed (blanks dropped, a frame equal to the previous
// FRAME dropped), mapped to code points, and scored against the reference transcripts:
//   CER numerator = Levenshtein over code points with ' ' removed from both, denominator = len(expected without ' ');
//   WER numerator = Levenshtein over the word sequences (str.split(): runs of Python whitespace), denominator = #words.
// totals[5] = {cer_err, cer_ref, wer_err, wer_ref, sum of len(decoded)}; hyp_cp / hyp_off (optional) receive the decoded
// code points, utterance n at hyp_cp[hyp_off[n] .. hyp_off[n+1]).
namespace {
inline bool py_isspace(int32_t c) {
    return (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x20) || c == 0x85 || c == 0xa0 || c == 0x1680 || (c >= 0x2000 && c <= 0x200a) ||
           c == 0x2028 || c == 0x2029 || c == 0x202f || c == 0x205f || c == 0x3000;
}
struct Word { const int32_t* p; int n; };
inline void split_words(const int32_t* s, int n, std::vector<Word>& out) {
    out.clear();
    int i = 0;
    while (i < n) {
        while (i < n && py_isspace(s[i])) ++i;
        const int b = i;
        while (i < n && !py_isspace(s[i])) ++i;
        if (i > b) out.push_back(Word{s + b, i - b});
    }
}
template <class Eq>
int edit_distance(int na, int nb, Eq eq, std::vector<int>& prev, std::vector<int>& cur) {
    prev.resize(nb + 1);
    cur.resize(nb + 1);
    for (int j = 0; j <= nb; ++j) prev[j] = j;
    for (int i = 1; i <= na; ++i) {
        cur[0] = i;
        for (int j = 1; j <= nb; ++j) cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + (eq(i - 1, j - 1) ? 0 : 1));
        std::swap(prev, cur);
    }
    return prev[nb];
}
}  // namespace

extern "C" int w2l_greedy_score_host(const int32_t* idx, int N, int T, const int32_t* sizes, int blank, const int32_t* lut,
                                     int n_labels, const int32_t* ref_cp, const int64_t* ref_off, int64_t* totals,
                                     int32_t* hyp_cp, int64_t* hyp_off) {
    W2L_CHECK_ARG(idx != nullptr && lut != nullptr && ref_cp != nullptr && ref_off != nullptr && totals != nullptr && N >= 0 && T >= 0,
                  "greedy_score_host: null pointer or negative size");
    std::vector<int32_t> hyp, a, b;
    std::vector<Word> wa, wb;
    std::vector<int> prev, cur;
    int64_t cer_err = 0, cer_ref = 0, wer_err = 0, wer_ref = 0, hyp_len = 0, written = 0;
    if (hyp_off != nullptr) hyp_off[0] = 0;
    for (int n = 0; n < N; ++n) {
        const int32_t* row = idx + (int64_t)n * T;
        int len = sizes != nullptr ? sizes[n] : T;
        len = std::max(0, std::min(len, T));
        hyp.clear();
        for (int t = 0; t < len; ++t) {
            const int32_t c = row[t];
            W2L_CHECK_ARG(c >= 0 && c < n_labels, "greedy_score_host: index %d outside the %d labels", (int)c, n_labels);
            if (c != blank && !(t > 0 && c == row[t - 1])) hyp.push_back(lut[c]);
        }
        hyp_len += (int64_t)hyp.size();
        if (hyp_cp != nullptr && hyp_off != nullptr) {
            std::copy(hyp.begin(), hyp.end(), hyp_cp + written);
            written += (int64_t)hyp.size();
            hyp_off[n + 1] = written;
        }
        const int32_t* ref = ref_cp + ref_off[n];
        const int nref = (int)(ref_off[n + 1] - ref_off[n]);
        a.clear();
        b.clear();
        for (int i = 0; i < nref; ++i) if (ref[i] != 0x20) a.push_back(ref[i]);
        for (int32_t c : hyp) if (c != 0x20) b.push_back(c);
        cer_err += edit_distance((int)a.size(), (int)b.size(), [&](int i, int j) { return a[i] == b[j]; }, prev, cur);
        cer_ref += (int64_t)a.size();
        split_words(ref, nref, wa);
        split_words(hyp.data(), (int)hyp.size(), wb);
        wer_err += edit_distance((int)wa.size(), (int)wb.size(), [&](int i, int j) {
            return wa[i].n == wb[j].n && memcmp(wa[i].p, wb[j].p, (size_t)wa[i].n * sizeof(int32_t)) == 0; }, prev, cur);
        wer_ref += (int64_t)wa.size();
    }
    totals[0] = cer_err; totals[1] = cer_ref; totals[2] = wer_err; totals[3] = wer_ref; totals[4] = hyp_len;
    return 0;
}

// ---- stream concurrency probe ------------------------------------------------------------------------------------
// Can a kernel launched on `stream_b` START while a large kernel launched earlier on `stream_a` is still being
// dispatched?  HIP maps a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES) and those onto the command
// processor's pipes; two streams that share one run their kernels one after the other -- a dispatch that does not fit the
// chip at once holds its queue until its last workgroup has been placed -- and the step loses the overlap of its side
// streams (measured: 20.0 instead of 14.0 ms/step when the RCCL communicator happened to be created before the weight-
// gradient stream).  Which streams collide depends on the creation order of EVERY stream in the process (torch's, RCCL's),
// so the host picks its side streams by measurement (wav2letter_pytorch_amd/streams.py).
//   fill:  `rounds` x (2 blocks per CU, held there by 64 KiB of LDS each), every block spins `spin_us`; stamps[0] = the first
//          block's start, stamps[1] = the last block's end (s_memrealtime ticks, 100 MHz, one counter chip-wide);
//   stamp: one wave, writes its start time to stamps[2].
// stamps: 3 x int64 on the device, ZERO and visible before the call (the caller synchronises); the caller synchronises
// again and reads them: (stamps[2] - stamps[0]) / (stamps[1] - stamps[0]) is ~0 when b runs beside a, ~1 when it queues.
namespace {
__global__ __launch_bounds__(256) void probe_fill_kernel(long long ticks, long long* stamps) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        lds[0] = 0;                                               // keep the allocation
        const long long t0 = wall_clock64();
        if (blockIdx.x == 0) stamps[0] = t0;
        // time advances, so every block leaves; the iteration cap (~2k cycles per s_sleep 32: a few ms) is the exit a
        // wave reaches even if the counter did not
        for (int it = 0; it < 4096 && wall_clock64() - t0 < ticks; ++it) __builtin_amdgcn_s_sleep(32);
        atomicMax((unsigned long long*)&stamps[1], (unsigned long long)wall_clock64());
    }
}
__global__ void probe_stamp_kernel(long long* stamps) {
    if (threadIdx.x == 0) stamps[2] = wall_clock64();
}
}  // namespace

extern "C" int w2l_stream_probe(void* stream_a, void* stream_b, void* stamps_dev, int rounds, int spin_us) {
    W2L_CHECK_ARG(stamps_dev != nullptr, "stream_probe: null pointer");
    W2L_CHECK_ARG(rounds >= 1 && rounds <= 64 && spin_us >= 1 && spin_us <= 1000, "stream_probe: rounds 1..64, spin 1..1000 us");
    int dev = 0, cus = 0;
    W2L_CHECK_HIP(hipGetDevice(&dev));
    W2L_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    W2L_CHECK_HIP(w2l_allow_big_lds((const void*)probe_fill_kernel));
    const long long ticks = (long long)spin_us * 100;            // s_memrealtime: 100 MHz
    hipLaunchKernelGGL(probe_fill_kernel, dim3(cus * 2 * rounds), dim3(256), 64 * 1024, (hipStream_t)stream_a, ticks,
                       (long long*)stamps_dev);
    W2L_CHECK_LAUNCH();
    hipLaunchKernelGGL(probe_stamp_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream_b, (long long*)stamps_dev);
    W2L_CHECK_LAUNCH();
    return 0;
}
