// Host side of the amdmsm C ABI (include/amdmsm.h): context, workspace, the MSM
// pipeline schedule and the host-buffer entry points.  All arithmetic happens in
// the kernels of msm_group.hip; there is no CPU arithmetic path in this library.
#include "../../include/amdmsm.h"
#include "engine_internal.h"
#include "group_vtable.h"

#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace amdmsm;

// One workspace "slot" per MSM in flight.  With pipeline depth > 1 consecutive
// amdmsm_msm_device calls take the slots round-robin, so calls issued on different streams
// may overlap on the device (the small-grid tail of one MSM under the bulk kernels of the
// next); a slot is reused only after the call that last used it has finished (event wait).
constexpr int MAX_SLOTS = 4;
// Window groups of one MSM (experimental, off by default: AMDMSM_WINDOW_GROUPS=2..4): the windows
// are accumulated group by group on the caller's stream, highest first; the tail of a finished
// group (bucket fix-up, reduction, its share of the Horner chain) runs on a side stream under
// the accumulation of the next group.  Measured gain is only ~3% at 2^20..2^22 and ~1% at 2^26:
// the tail kernels are short on parallelism, not on issue slots, and slow down 2-3x when they
// share SIMDs with k_accumulate.
constexpr int MAX_GROUPS = 4;
struct ws_slot {
    void *ws = nullptr;
    size_t ws_bytes = 0;
    hipEvent_t done = nullptr;
    bool used = false;
    hipStream_t side[MAX_GROUPS - 1] = {};
    hipEvent_t acc_done[MAX_GROUPS - 1] = {}, tail_done[MAX_GROUPS - 1] = {};
    long long last_ticket = -1;   // timing ticket of the call that last used the slot
    bool ev_valid = false;
    hipEvent_t ov_in = nullptr, ov_acc = nullptr, ov_tail = nullptr;   // overlap mode (see amdmsm_ctx::bulk_stream)
};

// grow-only device buffer kept by the context between calls (host-buffer entry points)
struct grow_buf {
    void *p = nullptr;
    size_t bytes = 0;
};

// Base vector imported once and kept in HBM as compact affine records (amdmsm_register_bases):
// the proving key of a prover that calls multi_exp with the same bases proof after proof.
// Host-buffer calls whose base range lies inside [host, host + n*stride) use the resident copy.
struct base_entry {
    uint64_t id = 0;
    int curve = 0, group = 0, form = 0;
    const char *host = nullptr;
    size_t n = 0, stride = 0;
    void *d_aff = nullptr;
    void *d_endo = nullptr;   // phi(P) records of the whole vector (endomorphism split), built at first use
    bool automatic = false;   // created by AMDMSM_BASE_CACHE, evictable
    bool pinned = false;      // resolved by the batch call that is running: not evictable until it returns
    uint64_t last_use = 0;
    size_t aff_bytes = 0;     // bytes of one compact affine record of this entry's group
    // HBM held by the entry: the affine copy and, once built, the phi(P) records
    size_t bytes() const { return n * aff_bytes * (d_endo ? 2 : 1); }
};

// fixed-base exponentiation state kept between amdmsm_batch_exp calls: buffers and the resident window table
struct fb_state {
    grow_buf small, table, table_aff, out;
    bool valid = false;
    int curve = 0, group = 0;
    size_t scalar_size = 0, window = 0;
    unsigned char g[3 * 24 * 2 * 4] = {};   // the generator the resident table was built from
};

// staging of the streaming entries (multi_exp_stream*), kept between calls
struct stream_state {
    static constexpr int NB = 2;
    void *h_stage[NB] = {};
    size_t h_bytes[NB] = {};
    grow_buf d_raw[NB], d_aff[NB], d_sc[NB], partials, status;
    hipStream_t streams[NB] = {};
};

struct amdmsm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    ws_slot slots[MAX_SLOTS];
    int depth = 1;
    unsigned next = 0;
    int last_slot = 0;
    bool timing = false;
    hipEvent_t ev[AMDMSM_MAX_PHASES + 1] = {};   // probes
    // phase events of the last TIMING_RING timed MSMs, indexed by ticket % TIMING_RING: a caller
    // can enqueue MSM after MSM without synchronising and read every call's phase times afterwards
    hipEvent_t ring[64][6] = {};
    uint64_t ticket = 0;      // tickets handed out so far (the last call's ticket is ticket - 1)
    void *chunk_partials = nullptr;               // a few points, for calls split into several MSMs
    // host-buffer entry points: staging in HBM reused across calls, a second stream that brings
    // the bases in while the scalars are already being sorted, and the resident base vectors
    grow_buf hb_src, hb_aff, hb_sc, hb_out, hb_stats;
    hipStream_t copy_stream = nullptr;
    hipEvent_t bases_ready = nullptr, host_done = nullptr;
    // Several MSMs in flight (pipeline depth > 1), overlap by construction: the bulk of every MSM -- bucket sort
    // and accumulation, which fill the device -- is enqueued on ONE stream in call order, its latency-bound tail
    // (bucket fix-up, reduction, Horner: a few waves, 0.6 ms of dependent chains at 2^20 points) on a second,
    // high-priority stream.  MSM k+1's sort and accumulation therefore start when MSM k's accumulation ends and
    // run under MSM k's tail; the accumulation kernel is launched one workgroup per CU short of full occupancy
    // in this mode so that a tail wave finds registers on every SIMD (group_vtable::accumulate_overlap).
    hipStream_t bulk_stream = nullptr, tail_stream = nullptr;
    std::vector<base_entry> bases;
    uint64_t next_base_id = 1, use_clock = 0;
    fb_state fb;
    stream_state ss;
    hipEvent_t aux_ev[5] = {};                    // phases of the last amdmsm_batch_exp
    float aux_ms[4] = {};
    std::string err;
    std::recursive_mutex mu;                      // one call at a time per context (entries nest)
};

namespace {

// The eight group translation units are linked weakly so a development build may carry a
// subset (AMDMSM_GROUPS=... python -m libff_amd.build); absent groups report UNSUPPORTED.
using vt_getter = const group_vtable *(*)();
const group_vtable *find_vt(int curve, int group) {
    static const vt_getter getters[] = {
        vt_alt_bn128_g1, vt_alt_bn128_g2, vt_bls12_377_g1, vt_bls12_377_g2, vt_bw6_761_g1, vt_bw6_761_g2,
        vt_bls12_381_g1, vt_bls12_381_g2,
    };
    for (vt_getter g : getters) {
        if (!g) continue;
        const group_vtable *v = g();
        if (v->curve == curve && v->group == group) return v;
    }
    return nullptr;
}

int fail(amdmsm_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) {                                                              \
            return fail(ctx, AMDMSM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                    \
    } while (0)

// libff::log2 (ceil), utils.cpp:32-44
size_t libff_log2(size_t n) {
    size_t r = ((n & (n - 1)) == 0 ? 0 : 1);
    while (n > 1) {
        n >>= 1;
        r++;
    }
    return r;
}

struct plan_t {
    int c = 0, W = 0;
    int D = 0;   // > 0: precomputed-table mode, D digits per scalar in one bucket set (W == 1)
    int G = 1;   // window groups (see ws_slot)
    size_t queue_stride = 0, off_partial = 0;
    uint32_t B = 0, L = 0;
    uint32_t S = 0, T = 0;   // entries per accumulation lane, lanes per window
    size_t off_counts = 0, off_lists = 0, off_buckets = 0, off_lvl0 = 0, off_lvl1 = 0, total = 0;
    size_t off_pfirst = 0, off_plast = 0, off_cont = 0, off_queue = 0;
    size_t off_coarse = 0, off_cursor = 0, off_tmp_payload = 0, off_tmp_key = 0, off_big = 0;
    // bucket reduction as row / column sums + bit planes (vt->reduce_rowcol): scratch and serial lengths
    bool rowcol = false;
    uint32_t q_row = 0, q_col = 0;
    size_t off_rc = 0, off_planes = 0, off_winsum = 0, rc_points = 0;
    int K = 1;                                  // MSMs the workspace holds side by side (amdmsm_msm_device_batch)
    size_t big_stride = 0, endo_stride = 0;     // bytes per MSM of the big-bin sort scratch / the phi(P) records
    size_t list_stride = 0;
    bool glv = false;        // endomorphism split: the sorted columns are 2 * n_real half-length scalars
    size_t off_endo = 0;     // phi(P) = (beta x, y) of every base, compact affine
};

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Windows needed for signed radix-2^c digits of the half scalars of the endomorphism split,
// |k| <= 2^lb: the recoded top digit must not carry out, i.e. 2^lb < 2^(cW-1) - 2^(c(W-1))
// (the "+ 2" of field_get_signed_digit's bits + 2, multiexp.tcc:584-586, with the real bound in
// place of the bit length: alt_bn128 |k| < 2^125.8 fits 8 windows of 16 bits).
int glv_windows(const group_vtable *vt, int c) {
    const double lb = vt->glv_bound_log2_x1000 / 1000.0 + 1e-3;
    int W = 1;
    while ((double)(c * W - 1) + std::log2(1.0 - std::ldexp(1.0, 1 - c)) < lb) ++W;
    return W;
}
int num_windows(const group_vtable *vt, int c, bool glv) { return glv ? glv_windows(vt, c) : (vt->fr_bits + 2 + c - 1) / c; }

// Window size.  Cost model fitted to measurements on MI355X (alt_bn128 G1, tools/sweep_c.py; the other
// fields scale every term alike, the wide ones pay more per bucket because their reduction
// kernels spill).  From 2^23 points up: sort + accumulation ~0.10 ns per (point, window) entry;
// fix-up + bucket reduction ~0.35 ms + 0.5 ns per bucket (0.98 ms for 16 x 2^16 buckets, 3.1 - 5.0
// for 13 x 2^19).  Below that the reduction is a latency floor (~0.65 ms from 2^16 points up) and
// c = 16 -- 2^15 buckets, exactly one reduction lane per SIMD lane -- is the measured optimum for
// every group from 2^16 to 2^22 points (c = 15 is slower even in the reduction: 0.74 vs 0.64 ms);
// the round-1 constants reproduce that and are kept there.  A top window that keeps only 2..8 significant bits concentrates all of
// its n entries in a handful of buckets, which the second sort level (one workgroup per coarse
// bin) processes almost serially -- such c are penalised rather than forbidden.
//
// (Round 3 re-sweep after the bucket reduction became plain sums, profiles/r03_sweep_c.txt: the choices below still hold --
// 2^16..2^21 c = 16 with the split (c = 13: 1.04 vs 0.68 ms at 2^16), 2^23 c = 17, 2^24..2^26 c = 20; bls12_377 G1 2^22 c = 17,
// bw6_761 G1 2^21 c = 16 (c = 14: 47.4 vs 45.2 ms), 2^24 c = 19; bls12_377 G2 2^21 c = 16 / 17, 2^24 c = 20.  A refit with
// per-field entry and bucket prices chose worse for the wide fields and was dropped.  Left on the table: alt_bn128 G1 2^22 with
// the split at c = 19 (7 windows) 7.16 vs 7.31-7.54 ms.)
// Modelled time (ns, alt_bn128 G1 scale) of an MSM over n points with window size c, with or without
// the endomorphism split.  The split doubles the digit columns and halves their length: about the
// same number of bucket entries, half the buckets to reduce and half the final doublings
// (~1.95 us each) -- measured on alt_bn128 G1 2^16 / 2^18 / 2^20 / 2^21 points: 1.02 vs 1.34,
// 1.34 vs 1.71, 2.66 vs 2.98, 4.38 vs 4.56 ms; its entries cost the same below 2^22 points and ~8 %
// more above, where bases plus phi(P) records (2 x 64 B x n) no longer fit the 256 MB Infinity
// Cache, so from 2^22 points up the plain path is level or ahead again (2^22: 7.78 vs 7.84,
// 2^23: 14.0 vs 13.7, 2^26: 87.4 vs 87.6 ms; profiles/r02_endomorphism_sweep.txt).
double plan_cost(const group_vtable *vt, size_t n, int c, bool glv) {
    const int bits = glv ? (vt->glv_bound_log2_x1000 + 999) / 1000 : vt->fr_bits;
    const int W = num_windows(vt, c, glv);
    const double cols = glv ? 2.0 * (double)n : (double)n;   // digit columns
    const double B = (double)((size_t)1 << (c - 1));
    // (24 limbs: the reduction kernels run one wave per SIMD; ~10 ns per bucket against 0.8 for 8 limbs,
    // while an accumulation entry costs 8.3x)
    const double wide = vt->fq_words >= 24 ? 2.2 : ((vt->fq_words > 8 || vt->el_words > vt->fq_words) ? 1.5 : 1.0);
    // (the wider fields hide the second base load behind their longer additions: same rate either way)
    // (round 3: with the cheaper reduction the split still wins a little above 2^22 points for the 8-limb field -- 2^22:
    // c = 19, 7 windows 7.16 ms against 7.3-7.5 plain; 2^23: 13.5 against 13.1 -- so its entries are priced level up to 5 M points;
    // with the reduced-radix accumulation: 2^23 12.0 either way, 2^24 21.8 (split, c = 19) against 22.8, 2^25 41.6 against 40.4,
    // 2^26 77.9 (c = 22) against 75.3 -- level up to 20 M points)
    const double entry = !glv ? 1.0 : (n < ((size_t)1 << 22) ? 1.0 : (wide > 1.0 ? 1.02 : (n < ((size_t)20 << 20) ? 1.0 : 1.08)));
    const bool large = n >= ((size_t)1 << 23);
    // windows that can hold a nonzero digit: with W * c well above the scalar length the top
    // window sees neither a scalar bit nor the carry (c = 17: 15 of 16 windows for a 254-bit
    // Fr) -- measured to pay from 2^23 points up (2^23: c = 17 14.4 ms vs c = 16 15.0 ms;
    // at 2^21 / 2^22 the accumulation time does not drop and c = 16 stays ahead)
    int W_work = W;
    if (large) {
        while (W_work > 1 && (W_work - 1) * c >= bits + 1) --W_work;
    }
    double cost;
    if (large) {
        cost = (double)W_work * cols * 0.10 * entry + 0.35e6 + (double)W * B * 0.5 * wide;
    } else {
        double reduce_cost = (double)W * B * 0.8 * wide;
        if (n >= 65536) reduce_cost = std::max(reduce_cost, 0.65e6);
        cost = (double)W * cols * 0.137 * entry + reduce_cost;
    }
    cost += (double)W * c * 1.95e3;   // final doublings
    const int top_bits = bits + 1 - (W - 1) * c;   // bit positions left for the top window
    if (top_bits >= 2 && top_bits <= 8) cost += cols * 3.0 / (double)(1 << (top_bits - 1));
    return cost;
}

// n: points
int choose_c(const group_vtable *vt, size_t n, bool glv = false, double *cost_out = nullptr) {
    if (n == 0) return 2;
    // Small and medium inputs are launch and chain latency, which the model does not describe; the choice there is measured
    // (profiles/r04_experiments.txt, every group, c = 2 .. 16 at 2^2 .. 2^17 points).  Since the bucket reduction became plain sums
    // (row / column sums, bit planes: c >= 10) the short Horner of few wide windows wins over few buckets: c = 10 up to 2^10
    // points (alt_bn128 G2 n = 4: 1.41 ms against 3.19 at the c = 2 the model used to pick, bw6_761 G1 2.1 against 3.4),
    // c = 13 -- a square 64 x 64 weight matrix -- from there to where c = 16 takes over: 2^15 points for the prime-field groups
    // under the endomorphism split, 2^17 for Fq2 and 761-bit coordinates, 2^18 without the split (alt_bn128 G1 2^12: 0.49 ms
    // against 0.61 at c = 8; alt_bn128 G2 2^16: 1.93 against 2.36 at c = 16).  Round 2's rule (c = 8 below 2^16) dated from
    // the segment kernels.
    {
        const bool slow_field = vt->fq_words >= 24 || vt->el_words > vt->fq_words;
        const size_t n13 = !glv ? ((size_t)1 << 18) : (slow_field ? ((size_t)1 << 17) : ((size_t)1 << 15));
        int small_c = 0;
        if (n < 1024) small_c = 10;
        else if (n < n13) small_c = (vt->fq_words >= 24 && n < ((size_t)1 << 15)) ? 12 : 13;
        else if (n < 65536) small_c = 16;   // 2^15 points under the split: 0.54 ms against 0.57 at c = 13 and 0.70 at c = 8
        if (small_c) {
            if (cost_out) *cost_out = plan_cost(vt, n, small_c, glv);
            return small_c;
        }
    }
    double best = 1e300;
    int best_c = 2;
    for (int c = 2; c <= 22; ++c) {
        const double cost = plan_cost(vt, n, c, glv);
        if (cost < best) {
            best = cost;
            best_c = c;
        }
    }
    if (cost_out) *cost_out = best;
    return best_c;
}

// table_digits > 0: every scalar contributes table_digits entries (one per digit, pointing at
// its precomputed multiple) to a single bucket set; n is then the number of ENTRIES.
// glv: n counts the 2 x points digit columns of the endomorphism split
// batch > 1: workspace for `batch` MSMs of the same shape side by side -- every per-window array holds batch * W windows, MSM j
// owning windows [j * W, (j + 1) * W), so that the tail kernels run once over all of them (amdmsm_msm_device_batch)
int make_plan(const group_vtable *vt, size_t n, int c_req, int L_req, plan_t &p, int S_req = 0, int table_digits = 0,
              int G_req = 0, bool glv = false, bool overlap = false, int batch = 1) {
    if (c_req < 0 || c_req > 24 || c_req == 1) return AMDMSM_ERR_BAD_ARG;
    if (table_digits && (c_req < 2 || c_req > 22)) return AMDMSM_ERR_BAD_ARG;
    if (glv && (table_digits || c_req > 22)) return AMDMSM_ERR_BAD_ARG;
    p.glv = glv;
    p.c = c_req ? c_req : choose_c(vt, glv ? n / 2 : n, glv);
    // field_get_signed_digit needs room for bits + 2 (multiexp.tcc:584-586)
    p.W = table_digits ? 1 : num_windows(vt, p.c, glv);
    p.D = table_digits;
    p.B = (uint32_t)1 << (p.c - 1);
    // buckets per reduction lane.  k_reduce_segments is bound by the dependent chain of one wave
    // (2 L additions + the segment-offset multiple + the 64:1 fold), so the best L is the one that
    // gives every SIMD about one wave: W * B / L ~ 65536 lanes (1024 waves), L in 2 .. 64.
    // Measured (alt_bn128 G1, reduction phase incl. fix-up): 2^20, c = 16: L = 4 / 8 / 16 ->
    // 0.86 / 0.66 / 0.84 ms; 2^23, c = 17: 8 / 16 / 32 -> 1.15 / 1.01 / 1.32; 2^26, c = 20:
    // 8 / 16 / 32 / 64 / 128 -> 6.80 / 5.43 / 5.09 / 5.09 / 5.62.
    // (groups whose reduction lanes are two physical lanes wide -- fp2h.cuh, reduce_fold == 32 --
    // need twice the L for the same wave count: bls12_377 G2 2^21 3.87 -> 2.96 ms with L = 16)
    const size_t red_lanes = (size_t)(64 / vt->reduce_fold);
    // The smallest L whose lanes the device holds at one wave per SIMD (65536, a few per cent more
    // still pays: bls12_381 G1, 17 windows of 2^15 buckets, L = 8 / 16 -> 69632 / 34816 lanes, 1.24 /
    // 1.53 ms), at most 64.  Window counts that are not powers of two used to land between one and
    // two waves per SIMD: bls12_377 G1 2^22 (15 x 2^16 buckets) L = 8 -> 16: 2.08 -> 1.83 ms,
    // bls12_377 G2 2^22 L = 16 -> 32: 5.36 -> 4.42 ms; bw6_761 G1 under the endomorphism split
    // (12 x 2^15) L = 4 / 8 / 16: 8.1 / 5.6 / 6.8 ms.
    uint32_t L = 2u;
    while (L < 64u && (size_t)p.W * (size_t)batch * p.B * red_lanes / L > (size_t)70000) L <<= 1;
    if (L_req > 0) L = (uint32_t)L_req;
    while (L > p.B) L >>= 1;
    if (L == 0 || (L & (L - 1))) return AMDMSM_ERR_BAD_ARG;
    p.L = L;
    const size_t xyz_bytes = (size_t)3 * vt->el_words * 4;
    p.list_stride = align_up(n ? n : 1, 64) + 16;   // + one chunk: a lane's last staged read may run past its entries
    // entries per lane S (<= 128, 256 in long launches; lane t of a window owns entries [t*S, (t+1)*S)).  All lanes do the
    // same work and the device holds `resident` of them at once, so the lane count W*T should
    // fill whole rounds: k rounds exactly for the smallest k that keeps S <= 128, or simply
    // S = 128 once there are many rounds anyway.
    {
        const double resident = (double)vt->accumulate_resident_lanes(overlap ? 1 : 0);
        const double entries = (double)n * p.W;
        uint32_t S = 8;
        if (entries >= 4.0 * 256.0 * resident) {
            // many rounds: longer lanes halve the partial records the fix-up and the export pass handle
            // (alt_bn128 G1, AMDMSM_ACC_S = 128 / 192 / 256 / 384 / 512: 2^26 75.5 / 74.8 / 74.0 / 74.1 / 74.6 ms,
            //  2^24 22.7 / 22.9 / 22.3 / 23.0 / 23.2 ms)
            S = 256;
        } else if (entries >= 8.0 * 128.0 * resident) {
            S = 128;
        } else if (entries > 8.0 * resident) {
            uint32_t k = 1;
            while (entries / (k * resident) > 128.0) ++k;
            S = (uint32_t)std::ceil(entries / (k * resident));
            while (S < 128 && (double)p.W * (double)((n + S - 1) / S) > k * resident) ++S;
        }
        if (S_req > 0) S = (uint32_t)S_req;
        p.S = S;
        p.T = (uint32_t)((n + S - 1) / S);
        if (p.T == 0) p.T = 1;
    }
    p.K = batch;
    const size_t Wt = (size_t)p.W * (size_t)batch;   // windows the workspace holds
    size_t off = 0;
    p.off_counts = off;
    off = align_up(off + Wt * p.B * 4, 256);
    p.off_lists = off;
    off = align_up(off + Wt * p.list_stride * 4, 256);
    const size_t zz_bytes = (size_t)vt->bucket_words * 4;   // (X, Y, ZZ, ZZZ) bucket accumulators (reduced-radix groups: 4 L limbs)
    p.off_buckets = off;
    off = align_up(off + Wt * p.B * zz_bytes, 256);
    const size_t M = p.B / p.L;
    p.off_lvl0 = off;
    off = align_up(off + Wt * M * xyz_bytes, 256);
    p.off_lvl1 = off;
    off = align_up(off + Wt * (M / 2 + 1) * xyz_bytes, 256);   // first fold leaves at most M / 32 points
    p.off_pfirst = off;
    off = align_up(off + Wt * p.T * zz_bytes, 256);
    p.off_plast = off;
    off = align_up(off + Wt * p.T * zz_bytes, 256);
    p.off_cont = off;
    off = align_up(off + Wt * p.T * 4, 256);
    p.G = G_req > 0 ? std::min(std::min(G_req, MAX_GROUPS), p.W) : 1;
    p.off_queue = off;   // one fix-up queue per group, each sized for the largest group
    p.queue_stride = align_up(fixup_queue_words((size_t)((p.W + p.G - 1) / p.G) * (size_t)batch * p.T) * 4, 256);
    off = off + p.queue_stride * p.G;
    p.off_partial = off;
    off = align_up(off + (size_t)MAX_GROUPS * xyz_bytes, 256);
    // two-level sort scratch
    // (the coarse counters and the header of the big-bin scratch are cleared by ONE memset: the
    // scratch follows the counters directly)
    p.off_coarse = off;
    off = align_up(off + Wt * 1025 * 4, 256);
    p.off_big = off;
    p.big_stride = p.c <= 22 ? align_up(sort_geometry(n, p.c, p.W).big_words * 4, 256) : 256;   // per MSM of a batch
    off += p.big_stride * (size_t)batch;
    p.off_cursor = off;
    off = align_up(off + Wt * 1024 * 4, 256);
    p.off_tmp_payload = off;
    off = align_up(off + Wt * p.list_stride * 4, 256);
    p.off_tmp_key = off;
    off = align_up(off + Wt * p.list_stride * 4, 256);   // 16-bit fine keys use half of it
    p.off_endo = off;
    p.endo_stride = glv ? align_up((n / 2) * (size_t)vt->el_words * 2 * 4, 256) : 0;   // per MSM of a batch
    off += p.endo_stride * (size_t)batch;
    // Bucket reduction as plain sums (row / column sums of the weight matrix, bit planes, one short
    // Horner per window) from c = 10 up; below that a window's few segments fold inside one wave of
    // k_reduce_segments.  AMDMSM_ROWCOL=0 keeps the segment kernels everywhere (A/B runs).
    // Lanes add q buckets serially before the butterfly: the smallest q (>= 4 / 4) that keeps the
    // row and column lanes together within about two waves per SIMD -- measured: see make_plan's note
    // in profiles/r03_experiments.txt.
    {
        static const int rc_env = getenv("AMDMSM_ROWCOL") ? atoi(getenv("AMDMSM_ROWCOL")) : 1;
        static const int rc_min_c = getenv("AMDMSM_ROWCOL_MIN_C") ? atoi(getenv("AMDMSM_ROWCOL_MIN_C")) : 10;
        static const int qr_env = getenv("AMDMSM_ROWCOL_QROW") ? atoi(getenv("AMDMSM_ROWCOL_QROW")) : 0;
        static const int qc_env = getenv("AMDMSM_ROWCOL_QCOL") ? atoi(getenv("AMDMSM_ROWCOL_QCOL")) : 0;
        p.rowcol = rc_env != 0 && p.c >= rc_min_c;
        const int h = p.c / 2;
        const size_t C = (size_t)1 << h, R = p.B >> h;
        // (measured, tools/exp_rowcol.sh: alt_bn128 G1 2^20 q = 4 / 8 / 16 -> reduction phase 0.46 / 0.51 / 0.51 ms, 2^23
        // 1.09 / 0.87 / 0.79; the wide fields, whose kernels hold one wave per SIMD, want one round of lanes:
        // bw6_761 G1 2^21 q = 8 / 16 / 32 -> 5.41 / 5.03 / 6.24 ms, bls12_377 G2 2^21 2.72 / 2.57 / 3.07)
        const bool one_wave = vt->fq_words >= 24 || vt->el_words > vt->fq_words;
        const size_t lane_budget = one_wave ? 70000 : 140000;
        uint32_t q = 4;
        while (q < 64 && Wt * p.B * red_lanes * 2 / q > lane_budget) q <<= 1;
        p.q_row = qr_env > 0 ? (uint32_t)qr_env : q;
        p.q_col = qc_env > 0 ? (uint32_t)qc_env : q;
        p.rc_points = R + 1 + C;
        p.off_rc = off;
        off = align_up(off + Wt * p.rc_points * xyz_bytes, 256);
        p.off_planes = off;
        off = align_up(off + Wt * p.c * xyz_bytes, 256);
        p.off_winsum = off;
        off = align_up(off + Wt * xyz_bytes, 256);
    }
    p.total = off;
    return AMDMSM_OK;
}

int ensure_ws(amdmsm_ctx *ctx, ws_slot &sl, size_t bytes) {
    if (sl.ws_bytes >= bytes) return AMDMSM_OK;
    if (sl.ws) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipFree(sl.ws));
        sl.ws = nullptr;
        sl.ws_bytes = 0;
    }
    const size_t want = bytes + bytes / 8;
    HIP_TRY(ctx, hipMalloc(&sl.ws, want));
    sl.ws_bytes = want;
    return AMDMSM_OK;
}

constexpr uint64_t TIMING_RING = 64;
void record(amdmsm_ctx *ctx, ws_slot &sl, int idx, hipStream_t st) {
    (void)sl;
    if (ctx->timing) (void)hipEventRecord(ctx->ring[ctx->ticket % TIMING_RING][idx], st);
}

// The whole single-GPU MSM on device-resident inputs.
// table_digits > 0: d_bases is a precompute_table with that many multiples per scalar
// (multi_exp_precompute_from_fifo, multiexp_stream.tcc:124-162: one bucket set, no doublings).
// before_accumulate (optional): called once the bucket sort has been enqueued -- the sort reads
// only the scalars, so a host entry uses it to bring the bases in on another stream meanwhile;
// it returns an event the accumulation must wait for (or null).
// The endomorphism split (msm_group.hip glv_split) computes k P as k1 P + k2 phi(P), which equals
// k P exactly where phi = [lambda]: on the order-r subgroup.  opts->endomorphism: 0 = permitted only
// for groups whose whole curve has order r (alt_bn128 G1: every base qualifies), 1 = permitted: the
// caller guarantees subgroup membership of every base (libff's G1 / G2 are that subgroup; the FFI
// entry points check it while decoding), 2 = the same guarantee, and use it whatever the size,
// -1 = never.  Where permitted it is used when the cost model favours it (plan_cost).
// AMDMSM_GLV=off switches it off everywhere; AMDMSM_GLV=on|force apply only where the caller left
// the option at 0 (experiments; logged once) -- a caller's -1 is never overridden.
bool use_endomorphism(const group_vtable *vt, size_t n, const amdmsm_opts *opts, int table_digits) {
    static const int env = [] {
        const char *e = getenv("AMDMSM_GLV");
        int v = 0;
        if (!e) return 0;
        if (!strcmp(e, "off") || !strcmp(e, "0")) v = -1;
        else if (!strcmp(e, "on") || !strcmp(e, "1")) v = 1;
        else if (!strcmp(e, "all") || !strcmp(e, "force") || !strcmp(e, "2")) v = 2;
        else if (!strcmp(e, "auto")) v = 3;
        if (v) fprintf(stderr, "[amdmsm] AMDMSM_GLV=%s overrides amdmsm_opts.endomorphism where the caller left it at 0 "
                               "(off: everywhere); on / force assert that every base lies in the order-r subgroup\n", e);
        return v;
    }();
    if (table_digits || n == 0 || n >= ((size_t)1 << 30)) return false;
    const int c_req = opts ? opts->window_bits : 0;
    if (c_req > 22) return false;
    int want = opts ? opts->endomorphism : 0;
    // The environment may switch the split off everywhere, and otherwise speaks only where the caller
    // expressed no choice (0): a caller's -1 (never) stays never, a caller's 1 / 2 keeps its meaning.
    if (env == -1) want = -1;
    else if ((env == 1 || env == 2) && want == 0) want = env;
    if (want < 0 || (want == 0 && !vt->prime_order)) return false;
    if (want >= 2) return true;
    // permitted: used where the model says it pays (small and medium inputs)
    double full = 0, split = 0;
    if (c_req) {
        full = plan_cost(vt, n, c_req, false);
        split = plan_cost(vt, n, c_req, true);
    } else {
        choose_c(vt, n, false, &full);
        choose_c(vt, n, true, &split);
    }
    return split < 0.98 * full;
}

int check_opts(amdmsm_ctx *ctx, const amdmsm_opts *opts) {
    if (opts && opts->struct_size != sizeof(amdmsm_opts))
        return fail(ctx, AMDMSM_ERR_BAD_ARG, "amdmsm_opts.struct_size does not match this library (AMDMSM_ABI_VERSION " +
                                                 std::to_string(AMDMSM_ABI_VERSION) + "): initialise with AMDMSM_OPTS_INIT");
    return AMDMSM_OK;
}
#define CHECK_OPTS(ctx, opts)                       \
    do {                                            \
        const int rc_opts_ = check_opts(ctx, opts); \
        if (rc_opts_) return rc_opts_;              \
    } while (0)

// One MSM handles at most this many points (its sorted lists index points with 31 bits, and the
// workspace grows with n: 42 GiB at 2^28 alt_bn128 points); longer inputs are cut into contiguous
// ranges whose partial results are summed, exactly the reference's chunk loop (multiexp.tcc:655-687).
// AMDMSM_MAX_RANGE_POINTS lowers the limit so that tests reach the split with small inputs.
constexpr size_t MAX_RANGES = 64;
size_t max_range_points() {
    static const size_t v = [] {
        const char *e = getenv("AMDMSM_MAX_RANGE_POINTS");
        const long long x = e ? atoll(e) : 0;
        return x > 0 ? (size_t)x : (size_t)1 << 28;
    }();
    return v;
}

struct msm_hook {
    int (*fn)(void *arg, hipEvent_t *wait_for) = nullptr;
    void *arg = nullptr;
};
int msm_device_impl(amdmsm_ctx *ctx, const group_vtable *vt, const uint32_t *d_bases, const uint32_t *d_scalars,
                    size_t n, uint32_t *d_out, const amdmsm_opts *opts, int table_digits = 0,
                    const msm_hook *hook = nullptr, const uint32_t *d_endo_resident = nullptr) {
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : ctx->stream;
    hipStream_t const user_st = st;
    const int form = opts ? opts->out_form : AMDMSM_OUT_LIBFF;
    const int mont = (opts && opts->scalars_plain) ? 0 : 1;
    static const bool atomic_sort = getenv("AMDMSM_SORT") && !strcmp(getenv("AMDMSM_SORT"), "atomic");
    const bool glv = use_endomorphism(vt, n, opts, table_digits) && !atomic_sort;
    const size_t entries = table_digits ? n * (size_t)table_digits : (glv ? 2 * n : n);   // per sorted list
    if (entries >= ((size_t)1 << 31)) return fail(ctx, AMDMSM_ERR_TOO_LARGE, "n (times table digits) must be < 2^31 per call");
    if (n == 0) {
        // empty sum = zero; sum_points over 0 points writes G::zero() in the requested form
        vt->sum_points(st, d_out, 0, form, d_out);
        HIP_TRY(ctx, hipGetLastError());
        return AMDMSM_OK;
    }
    plan_t p;
    // tuning knobs for experiments: AMDMSM_ACC_S (entries per accumulation lane)
    static const int acc_s_env = getenv("AMDMSM_ACC_S") ? atoi(getenv("AMDMSM_ACC_S")) : 0;
    static const int groups_env = getenv("AMDMSM_WINDOW_GROUPS") ? atoi(getenv("AMDMSM_WINDOW_GROUPS")) : 0;
    // overlap mode (several MSMs in flight, see amdmsm_ctx::bulk_stream): built and parity-tested, but OFF unless
    // AMDMSM_OVERLAP=1 -- measured on MI355X / ROCm 7.2 it loses (2^20, three in flight: 2.35 - 2.69 ms per MSM against
    // 2.09 - 2.14 with the ordering left to the hardware queues and 2.19 one at a time): the kernel traces in
    // profiles/r03_pipelined_*.txt show the intended schedule (sort and accumulation of MSM k+1 under the tail of MSM k)
    // but every kernel behind a cross-queue event wait starts ~50 us late, which outweighs the 0.7 ms tail it hides.
    static const bool overlap_env = getenv("AMDMSM_OVERLAP") && atoi(getenv("AMDMSM_OVERLAP")) != 0;
    const bool overlap = overlap_env && ctx->depth > 1 && ctx->bulk_stream && vt->accumulate_overlap_ok && !groups_env;
    int rc = make_plan(vt, entries, opts ? opts->window_bits : 0, opts ? opts->segment_len : 0, p, acc_s_env, table_digits,
                       groups_env, glv, overlap);
    if (rc) return fail(ctx, rc, "bad window_bits / segment_len");
    const int slot_idx = (int)(ctx->next++ % (unsigned)ctx->depth);
    ws_slot &sl = ctx->slots[slot_idx];
    ctx->last_slot = slot_idx;
    if (overlap) {
        // the bulk part runs on the context's bulk stream, behind whatever the caller's stream holds now
        HIP_TRY(ctx, hipEventRecord(sl.ov_in, user_st));
        st = ctx->bulk_stream;
        HIP_TRY(ctx, hipStreamWaitEvent(st, sl.ov_in, 0));
    }
    if (sl.used) HIP_TRY(ctx, hipStreamWaitEvent(st, sl.done, 0));   // previous user of this slot
    rc = ensure_ws(ctx, sl, p.total);
    if (rc) return rc;
    char *ws = (char *)sl.ws;
    uint32_t *counts = (uint32_t *)(ws + p.off_counts);
    uint32_t *lists = (uint32_t *)(ws + p.off_lists);
    uint32_t *buckets = (uint32_t *)(ws + p.off_buckets);
    uint32_t *lvl0 = (uint32_t *)(ws + p.off_lvl0);
    uint32_t *lvl1 = (uint32_t *)(ws + p.off_lvl1);

    record(ctx, sl, 0, st);
    // phi(P) of every base (endomorphism split): independent of the sort, so it runs beside it on
    // the slot's side stream when the bases are already on the device (no upload hook)
    // (d_endo_resident: the records already exist, kept with a registered base vector)
    const uint32_t *endo_pts = !glv ? nullptr : (d_endo_resident ? d_endo_resident : (const uint32_t *)(ws + p.off_endo));
    const bool endo_beside = glv && !d_endo_resident && !(hook && hook->fn);
    // the side stream also clears the bucket array and the queue heads while the sort runs (both
    // are first touched by the accumulation)
    HIP_TRY(ctx, hipEventRecord(sl.tail_done[0], st));
    if ((atomic_sort || p.c > 22) && !table_digits) {
        HIP_TRY(ctx, hipMemsetAsync(counts, 0, (size_t)p.W * p.B * 4, st));
        vt->count(st, d_scalars, n, mont, p.c, p.W, counts);
        record(ctx, sl, 1, st);
        vt->scatter(st, d_scalars, n, mont, p.c, p.W, counts, lists, p.list_stride);
    } else {
        HIP_TRY(ctx, hipMemsetAsync(ws + p.off_coarse, 0, p.off_big + 16 - p.off_coarse, st));   // counters + big-bin header
        record(ctx, sl, 1, st);
        vt->sort(st, d_scalars, n, mont, p.c, table_digits ? table_digits : p.W, (uint32_t *)(ws + p.off_coarse),
                 (uint32_t *)(ws + p.off_cursor), (int32_t *)lists, (uint32_t *)(ws + p.off_tmp_payload),
                 (uint32_t *)(ws + p.off_tmp_key), counts, lists, p.list_stride, (uint32_t *)(ws + p.off_big),
                 table_digits ? 1 : (glv ? 2 : 0), nullptr);
    }
    // (measured at 2^20 points: beside the whole sort 0.29 ms for the phase, beside its LDS-bound
    // second half only 0.35, after it 0.30; the plain path's sort takes 0.24)
    // enqueued after the sort kernels, ordered only behind the start of the call
    // (overlap mode keeps them on the bulk stream: the runtime multiplexes the many streams of a process over a few
    // hardware queues, and a side stream that lands in the queue of a caller's stream sits behind that stream's wait
    // for the previous MSM's tail -- seen in a kernel trace: the accumulation then started only after that tail)
    hipStream_t side = overlap ? st : sl.side[0];
    if (!overlap) HIP_TRY(ctx, hipStreamWaitEvent(side, sl.tail_done[0], 0));
    HIP_TRY(ctx, hipMemsetAsync(buckets, 0, (size_t)p.W * p.B * vt->bucket_words * 4, side));
    for (int g = 0; g < p.G; ++g) HIP_TRY(ctx, hipMemsetAsync(ws + p.off_queue + g * p.queue_stride, 0, 8, side));
    if (endo_beside) vt->endo_points(side, d_bases, n, (uint32_t *)(ws + p.off_endo));
    if (!overlap) HIP_TRY(ctx, hipEventRecord(sl.acc_done[0], side));
    if (hook && hook->fn) {
        hipEvent_t wait_for = nullptr;
        rc = hook->fn(hook->arg, &wait_for);
        if (rc) return rc;
        if (wait_for) HIP_TRY(ctx, hipStreamWaitEvent(st, wait_for, 0));
    }
    if (!overlap) HIP_TRY(ctx, hipStreamWaitEvent(st, sl.acc_done[0], 0));
    if (glv && !d_endo_resident && !endo_beside) vt->endo_points(st, d_bases, n, (uint32_t *)(ws + p.off_endo));
    const size_t zzw = (size_t)vt->bucket_words, xyzw = (size_t)vt->el_words * 3;   // words per XYZZ / Jacobian point
    const size_t M0 = p.B / p.L, cap1 = M0 / 2 + 1;
    uint32_t *pfirst = (uint32_t *)(ws + p.off_pfirst), *plast = (uint32_t *)(ws + p.off_plast);
    uint32_t *cont = (uint32_t *)(ws + p.off_cont), *partial = (uint32_t *)(ws + p.off_partial);
    // groups of windows, highest first: [w0, w0 + wg)
    int w_hi = p.W;
    for (int g = 0; g < p.G; ++g) {
        const int wg = p.W / p.G + (g < p.W % p.G ? 1 : 0);
        const int w0 = w_hi - wg;
        w_hi = w0;
        const bool last = g == p.G - 1;
        // accumulation group after group on the caller's stream; the tail of a finished group
        // moves to a side stream
        hipStream_t ts = last ? (overlap ? ctx->tail_stream : st) : sl.side[g];
        if (g == 0) record(ctx, sl, 2, st);
        vt->accumulate(st, counts + (size_t)w0 * p.B, lists + (size_t)w0 * p.list_stride, p.list_stride, d_bases,
                       buckets + (size_t)w0 * p.B * zzw, pfirst + (size_t)w0 * p.T * zzw, plast + (size_t)w0 * p.T * zzw,
                       cont + (size_t)w0 * p.T, wg, p.B, p.S, p.T, endo_pts, n, overlap ? 1 : 0);
        if (last) {
            record(ctx, sl, 3, st);
            if (overlap) {
                HIP_TRY(ctx, hipEventRecord(sl.ov_acc, st));
                HIP_TRY(ctx, hipStreamWaitEvent(ts, sl.ov_acc, 0));
            }
        } else {
            HIP_TRY(ctx, hipEventRecord(sl.acc_done[g], st));
            HIP_TRY(ctx, hipStreamWaitEvent(ts, sl.acc_done[g], 0));
        }
        vt->accumulate_fixup(ts, counts + (size_t)w0 * p.B, buckets + (size_t)w0 * p.B * zzw,
                             pfirst + (size_t)w0 * p.T * zzw, plast + (size_t)w0 * p.T * zzw, cont + (size_t)w0 * p.T,
                             (uint32_t *)(ws + p.off_queue + g * p.queue_stride), wg, p.B, p.S, p.T);
        uint32_t *src = lvl0 + (size_t)w0 * M0 * xyzw, *dst = lvl1 + (size_t)w0 * cap1 * xyzw;
        if (p.rowcol) {
            src = (uint32_t *)(ws + p.off_winsum) + (size_t)w0 * xyzw;
            vt->reduce_rowcol(ts, buckets + (size_t)w0 * p.B * zzw, wg, p.B, p.c, p.q_row, p.q_col,
                              (uint32_t *)(ws + p.off_rc) + (size_t)w0 * p.rc_points * xyzw,
                              (uint32_t *)(ws + p.off_planes) + (size_t)w0 * p.c * xyzw, src);
        } else {
            vt->reduce_segments(ts, buckets + (size_t)w0 * p.B * zzw, wg, p.B, p.L, src);
            const uint32_t fold = (uint32_t)vt->reduce_fold;
            uint32_t M = (uint32_t)M0;
            M /= std::min<uint32_t>(M, fold);   // folded per wave inside reduce_segments
            while (M > 1) {
                if ((size_t)M * (64 / fold) <= 256) {   // the remaining levels in one launch
                    vt->sum_block(ts, src, wg, M, dst);
                    M = 1;
                } else {
                    vt->sum_butterfly(ts, src, wg, M, dst);
                    M /= std::min<uint32_t>(M, fold);
                }
                std::swap(src, dst);
            }
        }
        // Horner over this group's windows, continuing from the groups above
        if (g > 0) HIP_TRY(ctx, hipStreamWaitEvent(ts, sl.tail_done[g - 1], 0));
        if (last) record(ctx, sl, 4, ts);
        vt->horner(ts, src, wg, p.c, last ? form : (int)AMDMSM_OUT_JACOBIAN, g > 0 ? partial + (size_t)(g - 1) * xyzw : nullptr,
                   last ? d_out : partial + (size_t)g * xyzw);
        if (!last) HIP_TRY(ctx, hipEventRecord(sl.tail_done[g], ts));
    }
    if (overlap) {
        // the result (and the slot) belong to the caller's stream again once the tail has run
        record(ctx, sl, 5, ctx->tail_stream);
        HIP_TRY(ctx, hipEventRecord(sl.ov_tail, ctx->tail_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(user_st, sl.ov_tail, 0));
        st = user_st;
    } else {
        record(ctx, sl, 5, st);
    }
    if (ctx->timing) sl.last_ticket = (long long)ctx->ticket++;
    sl.ev_valid = ctx->timing;
    HIP_TRY(ctx, hipEventRecord(sl.done, st));
    sl.used = true;
    HIP_TRY(ctx, hipGetLastError());
    return AMDMSM_OK;
}

// k MSMs of the same group and length in one call (amdmsm_msm_device_batch).  Sort and accumulation fill the device
// and run MSM after MSM; the tail -- fix-up, bucket reduction, final Horner: dependent chains of a few waves, a quarter
// of a 2^20-point MSM -- runs ONCE over the k * W windows of all of them, so its latency is paid once per batch
// instead of once per MSM (2^20 points, k = 3: see profiles/r03_experiments.txt).  This is the overlap a prover's
// back-to-back MSMs can really have on this device: in one stream, by making the latency-bound kernels wider.
constexpr int MAX_BATCH = 8;
int msm_device_batch_impl(amdmsm_ctx *ctx, const group_vtable *vt, int k, const uint32_t *const *d_bases,
                          const uint32_t *const *d_scalars, size_t n, uint32_t *const *d_outs, const amdmsm_opts *opts) {
    hipStream_t st = (opts && opts->stream) ? (hipStream_t)opts->stream : ctx->stream;
    const int form = opts ? opts->out_form : AMDMSM_OUT_LIBFF;
    const int mont = (opts && opts->scalars_plain) ? 0 : 1;
    const bool glv = use_endomorphism(vt, n, opts, 0);
    const size_t entries = glv ? 2 * n : n;
    plan_t p;
    int rc = make_plan(vt, entries, opts ? opts->window_bits : 0, opts ? opts->segment_len : 0, p, 0, 0, 0, glv, false, k);
    if (rc) return fail(ctx, rc, "bad window_bits / segment_len");
    if (p.c > 22) return fail(ctx, AMDMSM_ERR_BAD_ARG, "window_bits > 22 is not available for a batch");
    const int slot_idx = (int)(ctx->next++ % (unsigned)ctx->depth);
    ws_slot &sl = ctx->slots[slot_idx];
    ctx->last_slot = slot_idx;
    if (sl.used) HIP_TRY(ctx, hipStreamWaitEvent(st, sl.done, 0));
    rc = ensure_ws(ctx, sl, p.total);
    if (rc) return rc;
    char *ws = (char *)sl.ws;
    const size_t zzw = (size_t)vt->bucket_words, xyzw = (size_t)vt->el_words * 3;
    const size_t Wt = (size_t)p.W * (size_t)k;
    uint32_t *counts = (uint32_t *)(ws + p.off_counts), *lists = (uint32_t *)(ws + p.off_lists);
    uint32_t *buckets = (uint32_t *)(ws + p.off_buckets);
    uint32_t *pfirst = (uint32_t *)(ws + p.off_pfirst), *plast = (uint32_t *)(ws + p.off_plast), *cont = (uint32_t *)(ws + p.off_cont);
    record(ctx, sl, 0, st);
    HIP_TRY(ctx, hipMemsetAsync(ws + p.off_coarse, 0, Wt * 1025 * 4, st));
    for (int j = 0; j < k; ++j) HIP_TRY(ctx, hipMemsetAsync(ws + p.off_big + (size_t)j * p.big_stride, 0, 16, st));
    HIP_TRY(ctx, hipMemsetAsync(buckets, 0, Wt * p.B * vt->bucket_words * 4, st));
    HIP_TRY(ctx, hipMemsetAsync(ws + p.off_queue, 0, 8, st));
    record(ctx, sl, 1, st);
    for (int j = 0; j < k; ++j) {
        const size_t w0 = (size_t)j * p.W;
        uint32_t *lists_j = lists + w0 * p.list_stride, *counts_j = counts + w0 * p.B;
        vt->sort(st, d_scalars[j], n, mont, p.c, p.W, (uint32_t *)(ws + p.off_coarse) + w0 * 1025,
                 (uint32_t *)(ws + p.off_cursor) + w0 * 1024, (int32_t *)lists_j,
                 (uint32_t *)(ws + p.off_tmp_payload) + w0 * p.list_stride, (uint32_t *)(ws + p.off_tmp_key) + w0 * p.list_stride,
                 counts_j, lists_j, p.list_stride, (uint32_t *)(ws + p.off_big + (size_t)j * p.big_stride), glv ? 2 : 0, nullptr);
        uint32_t *endo_j = glv ? (uint32_t *)(ws + p.off_endo + (size_t)j * p.endo_stride) : nullptr;
        if (glv) vt->endo_points(st, d_bases[j], n, endo_j);
        if (j == 0) record(ctx, sl, 2, st);
        vt->accumulate(st, counts_j, lists_j, p.list_stride, d_bases[j], buckets + w0 * p.B * zzw, pfirst + w0 * p.T * zzw,
                       plast + w0 * p.T * zzw, cont + w0 * p.T, p.W, p.B, p.S, p.T, endo_j, n, 0);
    }
    record(ctx, sl, 3, st);
    vt->accumulate_fixup(st, counts, buckets, pfirst, plast, cont, (uint32_t *)(ws + p.off_queue), (int)Wt, p.B, p.S, p.T);
    uint32_t *src;
    if (p.rowcol) {
        src = (uint32_t *)(ws + p.off_winsum);
        vt->reduce_rowcol(st, buckets, (int)Wt, p.B, p.c, p.q_row, p.q_col, (uint32_t *)(ws + p.off_rc),
                          (uint32_t *)(ws + p.off_planes), src);
    } else {
        const size_t M0 = p.B / p.L, cap1 = M0 / 2 + 1;
        (void)cap1;
        src = (uint32_t *)(ws + p.off_lvl0);
        uint32_t *dst = (uint32_t *)(ws + p.off_lvl1);
        vt->reduce_segments(st, buckets, (int)Wt, p.B, p.L, src);
        const uint32_t fold = (uint32_t)vt->reduce_fold;
        uint32_t M = (uint32_t)M0;
        M /= std::min<uint32_t>(M, fold);
        while (M > 1) {
            if ((size_t)M * (64 / fold) <= 256) {
                vt->sum_block(st, src, (int)Wt, M, dst);
                M = 1;
            } else {
                vt->sum_butterfly(st, src, (int)Wt, M, dst);
                M /= std::min<uint32_t>(M, fold);
            }
            std::swap(src, dst);
        }
    }
    record(ctx, sl, 4, st);
    vt->horner_batch(st, src, k, p.W, p.c, form, d_outs);
    record(ctx, sl, 5, st);
    if (ctx->timing) sl.last_ticket = (long long)ctx->ticket++;
    sl.ev_valid = ctx->timing;
    HIP_TRY(ctx, hipEventRecord(sl.done, st));
    sl.used = true;
    HIP_TRY(ctx, hipGetLastError());
    return AMDMSM_OK;
}

int ensure_partials(amdmsm_ctx *ctx) {
    if (!ctx->chunk_partials) HIP_TRY(ctx, hipMalloc(&ctx->chunk_partials, MAX_RANGES * 3 * 24 * 2 * 4));
    return AMDMSM_OK;
}

// Device-resident MSM of any length: one msm_device_impl, or -- above max_range_points() -- the
// reference's chunk loop (multiexp.tcc:655-687: `one = total / chunks`, the last range takes the
// remainder, partial results summed) with ranges run one after the other on the caller's stream.
int msm_device_ranges(amdmsm_ctx *ctx, const group_vtable *vt, const uint32_t *d_bases, const uint32_t *d_scalars, size_t n,
                      uint32_t *d_out, const amdmsm_opts *opts) {
    const size_t maxr = max_range_points();
    if (n <= maxr) return msm_device_impl(ctx, vt, d_bases, d_scalars, n, d_out, opts);
    const size_t parts = (n + maxr - 1) / maxr;
    if (parts > MAX_RANGES) return fail(ctx, AMDMSM_ERR_TOO_LARGE, "input too large for one call");
    const size_t one = (n + parts - 1) / parts;   // <= maxr: no range exceeds the limit, the last one takes what is left
    const size_t xyz_bytes = (size_t)vt->el_words * 12, aff_bytes = (size_t)vt->el_words * 8, fr_bytes = (size_t)vt->fr_words * 4;
    int rc = ensure_partials(ctx);
    if (rc) return rc;
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    else o.out_form = AMDMSM_OUT_LIBFF;
    const int form = o.out_form;
    o.out_form = AMDMSM_OUT_JACOBIAN;
    hipStream_t st = o.stream ? (hipStream_t)o.stream : ctx->stream;
    for (size_t k = 0; k < parts; ++k) {
        const size_t lo = k * one, cnt = (k == parts - 1) ? n - lo : one;
        rc = msm_device_impl(ctx, vt, (const uint32_t *)((const char *)d_bases + lo * aff_bytes),
                             (const uint32_t *)((const char *)d_scalars + lo * fr_bytes), cnt,
                             (uint32_t *)((char *)ctx->chunk_partials + k * xyz_bytes), &o);
        if (rc) return rc;
    }
    vt->sum_points(st, (const uint32_t *)ctx->chunk_partials, (int)parts, form, d_out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));   // the partials buffer is shared by the context
    return AMDMSM_OK;
}

}  // namespace

const group_vtable *amdmsm_internal_find_vt(int curve, int group) { return find_vt(curve, group); }
void *amdmsm_internal_stream(amdmsm_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" {

int amdmsm_abi_version(void) { return AMDMSM_ABI_VERSION; }

int amdmsm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *amdmsm_strerror(int code) {
    switch (code) {
    case AMDMSM_OK: return "ok";
    case AMDMSM_ERR_NO_DEVICE: return "no gfx950 device available (this library has no CPU path)";
    case AMDMSM_ERR_BAD_ARG: return "bad argument";
    case AMDMSM_ERR_UNSUPPORTED: return "unsupported curve/group";
    case AMDMSM_ERR_HIP: return "HIP runtime error";
    case AMDMSM_ERR_TOO_LARGE: return "input too large for one call";
    default: return "unknown error";
    }
}

const char *amdmsm_last_error(const amdmsm_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int amdmsm_ctx_create(int device, amdmsm_ctx **out) {
    if (!out) return AMDMSM_ERR_BAD_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return AMDMSM_ERR_NO_DEVICE;
    if (device < 0 || device >= ndev) return AMDMSM_ERR_BAD_ARG;
    amdmsm_ctx *ctx = new amdmsm_ctx();
    ctx->device = device;
    dev_guard g(device);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return AMDMSM_ERR_HIP;
    }
    for (auto &e : ctx->ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            delete ctx;
            return AMDMSM_ERR_HIP;
        }
    }
    if (hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->bases_ready, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->host_done, hipEventDisableTiming) != hipSuccess) {
        delete ctx;
        return AMDMSM_ERR_HIP;
    }
    for (auto &set : ctx->ring) {
        for (auto &e : set) {
            if (hipEventCreate(&e) != hipSuccess) {
                delete ctx;
                return AMDMSM_ERR_HIP;
            }
        }
    }
    for (auto &e : ctx->aux_ev) {
        if (hipEventCreate(&e) != hipSuccess) {
            delete ctx;
            return AMDMSM_ERR_HIP;
        }
    }
    for (auto &sl : ctx->slots) {
        bool ok = hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) == hipSuccess;
        for (hipEvent_t *e : {&sl.ov_in, &sl.ov_acc, &sl.ov_tail}) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
        for (auto &e : sl.tail_done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto &e : sl.acc_done) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        for (auto &q : sl.side) ok = ok && hipStreamCreateWithFlags(&q, hipStreamNonBlocking) == hipSuccess;
        if (!ok) {
            delete ctx;
            return AMDMSM_ERR_HIP;
        }
    }
    *out = ctx;
    return AMDMSM_OK;
}

void amdmsm_ctx_destroy(amdmsm_ctx *ctx) {
    if (!ctx) return;
    {
        dev_guard g(ctx->device);
        (void)hipDeviceSynchronize();
        for (auto &sl : ctx->slots) {
            if (sl.ws) (void)hipFree(sl.ws);
            if (sl.done) (void)hipEventDestroy(sl.done);
            for (hipEvent_t e : {sl.ov_in, sl.ov_acc, sl.ov_tail}) {
                if (e) (void)hipEventDestroy(e);
            }
            for (auto &e : sl.acc_done) {
                if (e) (void)hipEventDestroy(e);
            }
            for (auto &e : sl.tail_done) {
                if (e) (void)hipEventDestroy(e);
            }
            for (auto &q : sl.side) {
                if (q) (void)hipStreamDestroy(q);
            }
        }
        for (auto &e : ctx->ev) {
            if (e) (void)hipEventDestroy(e);
        }
        for (auto &set : ctx->ring) {
            for (auto &e : set) {
                if (e) (void)hipEventDestroy(e);
            }
        }
        if (ctx->chunk_partials) (void)hipFree(ctx->chunk_partials);
        for (auto &e : ctx->aux_ev) {
            if (e) (void)hipEventDestroy(e);
        }
        for (int b = 0; b < stream_state::NB; ++b) {
            if (ctx->ss.h_stage[b]) (void)hipHostFree(ctx->ss.h_stage[b]);
            if (ctx->ss.streams[b]) (void)hipStreamDestroy(ctx->ss.streams[b]);
        }
        for (grow_buf *b : {&ctx->hb_src, &ctx->hb_aff, &ctx->hb_sc, &ctx->hb_out, &ctx->hb_stats, &ctx->fb.small, &ctx->fb.table,
                            &ctx->fb.table_aff, &ctx->fb.out, &ctx->ss.d_raw[0], &ctx->ss.d_raw[1], &ctx->ss.d_aff[0],
                            &ctx->ss.d_aff[1], &ctx->ss.d_sc[0], &ctx->ss.d_sc[1], &ctx->ss.partials, &ctx->ss.status}) {
            if (b->p) (void)hipFree(b->p);
        }
        for (auto &be : ctx->bases) {
            if (be.d_aff) (void)hipFree(be.d_aff);
            if (be.d_endo) (void)hipFree(be.d_endo);
        }
        if (ctx->bases_ready) (void)hipEventDestroy(ctx->bases_ready);
        if (ctx->host_done) (void)hipEventDestroy(ctx->host_done);
        if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
        if (ctx->bulk_stream) (void)hipStreamDestroy(ctx->bulk_stream);
        if (ctx->tail_stream) (void)hipStreamDestroy(ctx->tail_stream);
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
}

int amdmsm_sizes(int curve, int group, size_t out[4]) {
    const group_vtable *vt = find_vt(curve, group);
    if (!vt || !out) return AMDMSM_ERR_UNSUPPORTED;
    out[0] = (size_t)vt->fr_words * 4;
    out[1] = (size_t)vt->el_words * 3 * 4;
    out[2] = (size_t)vt->el_words * 2 * 4;
    out[3] = (size_t)vt->fr_bits;
    return AMDMSM_OK;
}

int amdmsm_plan_ex(int curve, int group, size_t n, int window_bits, int endomorphism, int *c, int *num_windows,
                   uint32_t *num_buckets, size_t *workspace_bytes, int *endomorphism_used) {
    const group_vtable *vt = find_vt(curve, group);
    if (!vt) return AMDMSM_ERR_UNSUPPORTED;
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    o.window_bits = window_bits;
    o.endomorphism = endomorphism;
    const bool glv = use_endomorphism(vt, n, &o, 0);
    plan_t p;
    const int rc = make_plan(vt, glv ? 2 * n : n, window_bits, 0, p, 0, 0, 0, glv);
    if (rc) return rc;
    if (c) *c = p.c;
    if (num_windows) *num_windows = p.W;
    if (num_buckets) *num_buckets = p.B;
    if (workspace_bytes) *workspace_bytes = p.total;
    if (endomorphism_used) *endomorphism_used = glv ? 1 : 0;
    return AMDMSM_OK;
}

int amdmsm_plan(int curve, int group, size_t n, int window_bits, int *c, int *num_windows, uint32_t *num_buckets,
                size_t *workspace_bytes) {
    return amdmsm_plan_ex(curve, group, n, window_bits, 0, c, num_windows, num_buckets, workspace_bytes, nullptr);
}

size_t amdmsm_pippenger_optimal_c(size_t num_elements) {
    // multiexp.tcc:35-40 (size_t wrap-around arithmetic kept as is)
    const size_t l = libff_log2(num_elements);
    return l - (l / 3 - 2);
}

size_t amdmsm_bdlo12_signed_optimal_c(size_t num_elements) {
    return amdmsm_pippenger_optimal_c(num_elements) + 1;   // multiexp.tcc:637-641
}

int amdmsm_set_timing(amdmsm_ctx *ctx, int enable) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    ctx->timing = enable != 0;
    for (auto &sl : ctx->slots) sl.ev_valid = false;
    return AMDMSM_OK;
}

int amdmsm_set_pipeline_depth(amdmsm_ctx *ctx, int depth) {
    if (!ctx || depth < 1 || depth > MAX_SLOTS) return AMDMSM_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    static const bool overlap_wanted = getenv("AMDMSM_OVERLAP") && atoi(getenv("AMDMSM_OVERLAP")) != 0;
    if (depth > 1 && overlap_wanted && !ctx->bulk_stream) {
        // overlap mode's two streams, created when first needed.
        // Both streams need hardware queues of their own: the runtime multiplexes a process's streams over a few
        // hardware queues, and a stream that shares one with a caller's stream sits behind that stream's barrier
        // packets (its wait for the previous MSM's tail) -- seen in kernel traces as a bulk part that starts only after
        // that tail.  Streams created with a CU mask get a dedicated queue (the mask is a queue property); the mask
        // here names every CU.  AMDMSM_STREAM_KIND: 0 plain streams, 1 tails on a high-priority stream, 2 (default)
        // full-CU-mask streams, 3 both high priority (experiments, profiles/r03_experiments.txt).
        static const int kind = getenv("AMDMSM_STREAM_KIND") ? atoi(getenv("AMDMSM_STREAM_KIND")) : 2;
        int least = 0, greatest = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (kind == 2) {
            int cus = 256;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device);
            std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0xffffffffu);
            if (cus % 32) mask.back() = (1u << (cus % 32)) - 1u;
            HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&ctx->bulk_stream, (uint32_t)mask.size(), mask.data()));
            HIP_TRY(ctx, hipExtStreamCreateWithCUMask(&ctx->tail_stream, (uint32_t)mask.size(), mask.data()));
        } else {
            HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->bulk_stream, hipStreamNonBlocking, kind == 3 ? greatest : 0));
            HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->tail_stream, hipStreamNonBlocking, kind == 0 ? 0 : greatest));
        }
    }
    ctx->depth = depth;
    ctx->next = 0;
    return AMDMSM_OK;
}

int amdmsm_last_slot(amdmsm_ctx *ctx) { return ctx ? ctx->last_slot : AMDMSM_ERR_BAD_ARG; }

int amdmsm_get_timings_by_ticket(amdmsm_ctx *ctx, long long ticket, float ms[AMDMSM_MAX_PHASES]);

int amdmsm_get_slot_timings(amdmsm_ctx *ctx, int slot, float ms[AMDMSM_MAX_PHASES]) {
    if (!ctx || !ms || slot < 0 || slot >= MAX_SLOTS) return AMDMSM_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    ws_slot &sl = ctx->slots[slot];
    if (!sl.ev_valid) return fail(ctx, AMDMSM_ERR_BAD_ARG, "no timed amdmsm_msm_device call recorded in this slot");
    return amdmsm_get_timings_by_ticket(ctx, sl.last_ticket, ms);
}

long long amdmsm_last_timing_ticket(amdmsm_ctx *ctx) {
    if (!ctx) return -1;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    if (!ctx->timing || ctx->ticket == 0) return -1;
    return (long long)(ctx->ticket - 1);
}

int amdmsm_get_timings_by_ticket(amdmsm_ctx *ctx, long long ticket, float ms[AMDMSM_MAX_PHASES]) {
    if (!ctx || !ms || ticket < 0) return AMDMSM_ERR_BAD_ARG;
    // under the context lock: no MSM of another thread re-records ring events meanwhile.  The slot of
    // ticket (next - TIMING_RING) is the one the next (or a failed) call records into: not readable.
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    for (int i = 0; i < AMDMSM_MAX_PHASES; ++i) ms[i] = 0.f;
    if ((uint64_t)ticket >= ctx->ticket || ctx->ticket - (uint64_t)ticket >= TIMING_RING)
        return fail(ctx, AMDMSM_ERR_BAD_ARG, "timing ticket is not (or no longer) held");
    hipEvent_t *ev = ctx->ring[(uint64_t)ticket % TIMING_RING];
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipEventSynchronize(ev[5]));
    for (int i = 0; i < 5; ++i) HIP_TRY(ctx, hipEventElapsedTime(&ms[i], ev[i], ev[i + 1]));
    HIP_TRY(ctx, hipEventElapsedTime(&ms[AMDMSM_PH_TOTAL], ev[0], ev[5]));
    return AMDMSM_OK;
}

int amdmsm_get_timings(amdmsm_ctx *ctx, float ms[AMDMSM_MAX_PHASES]) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    return amdmsm_get_slot_timings(ctx, ctx->last_slot, ms);
}

#define GET_VT(ctx, curve, group)                                         \
    if (!ctx) return AMDMSM_ERR_BAD_ARG;                                  \
    const group_vtable *vt = find_vt(curve, group);                       \
    if (!vt) return fail(ctx, AMDMSM_ERR_UNSUPPORTED, "unknown curve/group"); \
    std::lock_guard<std::recursive_mutex> lock_(ctx->mu);                           \
    dev_guard guard_(ctx->device)

int amdmsm_msm_device(amdmsm_ctx *ctx, int curve, int group, const void *d_bases_affine, const void *d_scalars,
                      size_t n, void *d_out_xyz, const amdmsm_opts *opts) {
    GET_VT(ctx, curve, group);
    CHECK_OPTS(ctx, opts);
    if (!d_out_xyz || (n && (!d_bases_affine || !d_scalars))) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
    return msm_device_ranges(ctx, vt, (const uint32_t *)d_bases_affine, (const uint32_t *)d_scalars, n,
                             (uint32_t *)d_out_xyz, opts);
}

int amdmsm_msm_device_batch(amdmsm_ctx *ctx, int curve, int group, int k, const void *const *d_bases_affine,
                            const void *const *d_scalars, size_t n, void *const *d_out_xyz, const amdmsm_opts *opts) {
    GET_VT(ctx, curve, group);
    CHECK_OPTS(ctx, opts);
    if (k < 1 || k > MAX_BATCH || !d_bases_affine || !d_scalars || !d_out_xyz) return fail(ctx, AMDMSM_ERR_BAD_ARG, "batch of 1 .. 8 MSMs");
    for (int j = 0; j < k; ++j) {
        if (!d_out_xyz[j] || (n && (!d_bases_affine[j] || !d_scalars[j]))) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
    }
    // one MSM, an empty one, or one too long for a single pass: the MSMs one after the other
    if (k == 1 || n == 0 || n > max_range_points() || (size_t)k * n >= ((size_t)1 << 30)) {
        for (int j = 0; j < k; ++j) {
            const int rc = msm_device_ranges(ctx, vt, (const uint32_t *)d_bases_affine[j], (const uint32_t *)d_scalars[j], n,
                                             (uint32_t *)d_out_xyz[j], opts);
            if (rc) return rc;
        }
        return AMDMSM_OK;
    }
    return msm_device_batch_impl(ctx, vt, k, (const uint32_t *const *)d_bases_affine, (const uint32_t *const *)d_scalars, n,
                                 (uint32_t *const *)d_out_xyz, opts);
}

size_t amdmsm_precompute_num_digits(int curve, size_t c) {
    // multiexp_stream.tcc:205, profile_multiexp.cpp:126: (FieldT::num_bits + c - 1) / c
    const group_vtable *vt = find_vt(curve, AMDMSM_G1);
    if (!vt || c == 0) return 0;
    return ((size_t)vt->fr_bits + c - 1) / c;
}

int amdmsm_precompute_bases_device(amdmsm_ctx *ctx, int curve, int group, const void *d_bases_affine, size_t n,
                                   size_t c, size_t num_digits, void *d_table, void *stream) {
    GET_VT(ctx, curve, group);
    if (c < 2 || c > 22 || num_digits < 1 || num_digits > 512) return fail(ctx, AMDMSM_ERR_BAD_ARG, "c / num_digits");
    if (n && (!d_bases_affine || !d_table)) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
    if (!n) return AMDMSM_OK;
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const size_t aff_bytes = (size_t)vt->el_words * 8;
    const size_t chunk = std::min(n, (size_t)1 << 18);
    void *tmp = nullptr;
    HIP_TRY(ctx, hipMalloc(&tmp, chunk * num_digits * aff_bytes));
    for (size_t lo = 0; lo < n; lo += chunk) {
        const size_t cnt = std::min(chunk, n - lo);
        vt->precompute_table(st, (const uint32_t *)((const char *)d_bases_affine + lo * aff_bytes), cnt, (int)c,
                             (int)num_digits, (uint32_t *)tmp, (uint32_t *)((char *)d_table + lo * num_digits * aff_bytes));
    }
    const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    HIP_TRY(ctx, e1);
    HIP_TRY(ctx, e2);
    return AMDMSM_OK;
}

int amdmsm_msm_precomputed_device(amdmsm_ctx *ctx, int curve, int group, const void *d_table, const void *d_scalars,
                                  size_t n, size_t c, size_t num_digits, void *d_out_xyz, const amdmsm_opts *opts) {
    GET_VT(ctx, curve, group);
    CHECK_OPTS(ctx, opts);
    if (!d_out_xyz || (n && (!d_table || !d_scalars))) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
    if (c < 2 || c > 22 || num_digits < 1 || num_digits > 512) return fail(ctx, AMDMSM_ERR_BAD_ARG, "c / num_digits");
    // more digits than the scalar has windows (+1 for the final carry) would only index multiples
    // that are never selected; such a table layout is a caller error
    if (num_digits > ((size_t)vt->fr_bits + c - 1) / c + 1) return fail(ctx, AMDMSM_ERR_BAD_ARG, "num_digits exceeds ceil(bits/c) + 1");
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    else o.out_form = AMDMSM_OUT_LIBFF;
    o.window_bits = (int)c;
    // one sorted list holds n * num_digits entries and is indexed with 31 bits: larger inputs are
    // split into ranges of points whose partial results are summed (multiexp.tcc:663-687 shape)
    // (AMDMSM_TABLE_MAX_ENTRIES lowers the limit so that tests can reach the split path)
    static const size_t max_entries = getenv("AMDMSM_TABLE_MAX_ENTRIES") ? (size_t)atoll(getenv("AMDMSM_TABLE_MAX_ENTRIES"))
                                                                         : ((size_t)1 << 31) - 1;
    const size_t max_pts = std::max<size_t>(1, max_entries / num_digits);
    if (n <= max_pts) {
        return msm_device_impl(ctx, vt, (const uint32_t *)d_table, (const uint32_t *)d_scalars, n, (uint32_t *)d_out_xyz,
                               &o, (int)num_digits);
    }
    const size_t parts = (n + max_pts - 1) / max_pts;
    if (parts > MAX_RANGES) return fail(ctx, AMDMSM_ERR_TOO_LARGE, "input too large for one call");
    const size_t xyz_bytes = (size_t)vt->el_words * 12, aff_bytes = (size_t)vt->el_words * 8;
    {
        const int rcp = ensure_partials(ctx);
        if (rcp) return rcp;
    }
    const int form = o.out_form;
    o.out_form = AMDMSM_OUT_JACOBIAN;
    hipStream_t st = o.stream ? (hipStream_t)o.stream : ctx->stream;
    for (size_t k = 0; k < parts; ++k) {
        const size_t lo = k * max_pts, cnt = std::min(max_pts, n - lo);
        const int rc = msm_device_impl(ctx, vt, (const uint32_t *)((const char *)d_table + lo * num_digits * aff_bytes),
                                       (const uint32_t *)((const char *)d_scalars + lo * (size_t)vt->fr_words * 4), cnt,
                                       (uint32_t *)((char *)ctx->chunk_partials + k * xyz_bytes), &o, (int)num_digits);
        if (rc) return rc;
    }
    vt->sum_points(st, (const uint32_t *)ctx->chunk_partials, (int)parts, form, (uint32_t *)d_out_xyz);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));   // the partials buffer is shared by the context
    return AMDMSM_OK;
}

int amdmsm_import_bases_device(amdmsm_ctx *ctx, int curve, int group, const void *d_src_xyz, size_t stride_bytes,
                               int base_form, size_t n, void *d_dst_affine, void *stream) {
    GET_VT(ctx, curve, group);
    if (stride_bytes % 16 || stride_bytes < (size_t)vt->el_words * 12) return fail(ctx, AMDMSM_ERR_BAD_ARG, "stride");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    vt->import_bases(st, (const uint32_t *)d_src_xyz, stride_bytes / 4, base_form == AMDMSM_FORM_SPECIAL, n,
                     (uint32_t *)d_dst_affine);
    HIP_TRY(ctx, hipGetLastError());
    return AMDMSM_OK;
}

int amdmsm_export_affine_device(amdmsm_ctx *ctx, int curve, int group, const void *d_src_affine, size_t n,
                                void *d_dst_xyz, void *stream) {
    GET_VT(ctx, curve, group);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    vt->export_affine(st, (const uint32_t *)d_src_affine, n, (uint32_t *)d_dst_xyz);
    HIP_TRY(ctx, hipGetLastError());
    return AMDMSM_OK;
}

int amdmsm_sum_points_device(amdmsm_ctx *ctx, int curve, int group, const void *d_points_jacobian, int k,
                             int out_form, void *d_out_xyz, void *stream) {
    GET_VT(ctx, curve, group);
    if (k < 0 || !d_out_xyz) return fail(ctx, AMDMSM_ERR_BAD_ARG, "k");
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    vt->sum_points(st, (const uint32_t *)d_points_jacobian, k, out_form, (uint32_t *)d_out_xyz);
    HIP_TRY(ctx, hipGetLastError());
    return AMDMSM_OK;
}

int amdmsm_gen_bases_seq_device(amdmsm_ctx *ctx, int curve, int group, uint64_t first, size_t n, void *d_dst_affine,
                                void *stream) {
    GET_VT(ctx, curve, group);
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    vt->gen_bases_seq(st, first, n, (uint32_t *)d_dst_affine);
    HIP_TRY(ctx, hipGetLastError());
    return AMDMSM_OK;
}

int amdmsm_field_op_device(amdmsm_ctx *ctx, int curve, int group, int op, const void *d_a, const void *d_b,
                           void *d_out, size_t n) {
    GET_VT(ctx, curve, group);
    vt->field_op(ctx->stream, op, (const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_out, n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AMDMSM_OK;
}

int amdmsm_group_op_device(amdmsm_ctx *ctx, int curve, int group, int op, const void *d_a, const void *d_b,
                           void *d_out, size_t n, int out_form) {
    GET_VT(ctx, curve, group);
    vt->group_op(ctx->stream, op, (const uint32_t *)d_a, (const uint32_t *)d_b, (uint32_t *)d_out, n, out_form);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AMDMSM_OK;
}

int amdmsm_digits_device(amdmsm_ctx *ctx, int curve, int group, const void *d_scalars, size_t n, int scalars_plain,
                         int c, int num_windows, int32_t *d_out) {
    GET_VT(ctx, curve, group);
    if (c < 2 || c > 24 || num_windows < 1) return fail(ctx, AMDMSM_ERR_BAD_ARG, "c / num_windows");
    vt->digits(ctx->stream, (const uint32_t *)d_scalars, n, scalars_plain ? 0 : 1, c, num_windows, d_out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AMDMSM_OK;
}

int amdmsm_endomorphism_digits_device(amdmsm_ctx *ctx, int curve, int group, const void *d_scalars, size_t n,
                                      int scalars_plain, int c, int num_windows, int32_t *d_out) {
    GET_VT(ctx, curve, group);
    if (c < 2 || c > 24 || num_windows < 1) return fail(ctx, AMDMSM_ERR_BAD_ARG, "c / num_windows");
    vt->glv_digits(ctx->stream, (const uint32_t *)d_scalars, n, scalars_plain ? 0 : 1, c, num_windows, d_out);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AMDMSM_OK;
}

int amdmsm_endomorphism_info(int curve, int group, void *lambda_plain, int *bound_log2_x1000, int *prime_order) {
    const group_vtable *vt = find_vt(curve, group);
    if (!vt) return AMDMSM_ERR_UNSUPPORTED;
    if (lambda_plain) memcpy(lambda_plain, vt->glv_lambda, (size_t)vt->fr_words * 4);
    if (bound_log2_x1000) *bound_log2_x1000 = vt->glv_bound_log2_x1000;
    if (prime_order) *prime_order = vt->prime_order;
    return AMDMSM_OK;
}

int amdmsm_mul_bench_device(amdmsm_ctx *ctx, int curve, int group, void *d_inout, size_t nthreads, int iters,
                            int inline_variant, float *ms) {
    GET_VT(ctx, curve, group);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[AMDMSM_MAX_PHASES - 1], ctx->stream));
    vt->mul_bench(ctx->stream, (uint32_t *)d_inout, nthreads, iters, inline_variant);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[AMDMSM_MAX_PHASES], ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[AMDMSM_MAX_PHASES]));
    if (ms) HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev[AMDMSM_MAX_PHASES - 1], ctx->ev[AMDMSM_MAX_PHASES]));
    return AMDMSM_OK;
}

int amdmsm_madd_bench_device(amdmsm_ctx *ctx, int curve, int group, const void *d_points_affine, void *d_out_xyz,
                             size_t nthreads, int iters, int inline_variant, float *ms) {
    GET_VT(ctx, curve, group);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[AMDMSM_MAX_PHASES - 1], ctx->stream));
    vt->madd_bench(ctx->stream, (const uint32_t *)d_points_affine, (uint32_t *)d_out_xyz, nthreads, iters,
                   inline_variant);
    HIP_TRY(ctx, hipEventRecord(ctx->ev[AMDMSM_MAX_PHASES], ctx->stream));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[AMDMSM_MAX_PHASES]));
    if (ms) HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev[AMDMSM_MAX_PHASES - 1], ctx->ev[AMDMSM_MAX_PHASES]));
    return AMDMSM_OK;
}

int amdmsm_malloc(amdmsm_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return AMDMSM_ERR_BAD_ARG;
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipMalloc(d_ptr, bytes ? bytes : 16));
    return AMDMSM_OK;
}

int amdmsm_free(amdmsm_ctx *ctx, void *d_ptr) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipFree(d_ptr));
    return AMDMSM_OK;
}

int amdmsm_memcpy_h2d(amdmsm_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AMDMSM_OK;
}

int amdmsm_memcpy_d2h(amdmsm_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return AMDMSM_OK;
}

int amdmsm_synchronize(amdmsm_ctx *ctx) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    return AMDMSM_OK;
}

// ------------------------------------------------------------ host entries
}   // extern "C"

namespace {

int ensure_buf(amdmsm_ctx *ctx, grow_buf &b, size_t bytes) {
    if (b.p && b.bytes >= bytes) return AMDMSM_OK;
    if (b.p) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        HIP_TRY(ctx, hipFree(b.p));
        b.p = nullptr;
        b.bytes = 0;
    }
    const size_t want = std::max<size_t>(bytes + bytes / 8, 256);
    HIP_TRY(ctx, hipMalloc(&b.p, want));
    b.bytes = want;
    return AMDMSM_OK;
}

// resident compact-affine copy of the host range [bases, bases + n*stride), or null
void *find_resident_bases(amdmsm_ctx *ctx, const group_vtable *vt, const void *bases, size_t stride, int form, size_t n,
                          base_entry **entry = nullptr, size_t *first = nullptr) {
    const char *lo = (const char *)bases;
    for (auto &e : ctx->bases) {
        if (e.curve != vt->curve || e.group != vt->group || e.form != form || e.stride != stride) continue;
        if (lo < e.host || lo + n * stride > e.host + e.n * e.stride) continue;
        const size_t off = (size_t)(lo - e.host);
        if (off % stride) continue;
        e.last_use = ++ctx->use_clock;
        if (entry) *entry = &e;
        if (first) *first = off / stride;
        return (char *)e.d_aff + (off / stride) * (size_t)vt->el_words * 8;
    }
    return nullptr;
}

int register_bases_impl(amdmsm_ctx *ctx, const group_vtable *vt, const void *bases, size_t stride, int form, size_t n,
                        bool automatic, uint64_t *handle) {
    const size_t aff_bytes = (size_t)vt->el_words * 8;
    base_entry e;
    e.curve = vt->curve;
    e.group = vt->group;
    e.form = form;
    e.host = (const char *)bases;
    e.n = n;
    e.stride = stride;
    e.automatic = automatic;
    e.aff_bytes = aff_bytes;
    HIP_TRY(ctx, hipMalloc(&e.d_aff, std::max<size_t>(n * aff_bytes, 256)));
    int rc = ensure_buf(ctx, ctx->hb_src, n * stride);
    if (rc == AMDMSM_OK) {
        hipError_t he = hipMemcpyAsync(ctx->hb_src.p, bases, n * stride, hipMemcpyHostToDevice, ctx->stream);
        if (he == hipSuccess) {
            vt->import_bases(ctx->stream, (const uint32_t *)ctx->hb_src.p, stride / 4, form == AMDMSM_FORM_SPECIAL, n,
                             (uint32_t *)e.d_aff);
            he = hipGetLastError();
        }
        if (he == hipSuccess) he = hipStreamSynchronize(ctx->stream);
        if (he != hipSuccess) rc = fail(ctx, AMDMSM_ERR_HIP, std::string("register_bases: ") + hipGetErrorString(he));
    }
    if (rc) {
        (void)hipFree(e.d_aff);
        return rc;
    }
    e.id = ctx->next_base_id++;
    e.last_use = ++ctx->use_clock;
    ctx->bases.push_back(e);
    if (handle) *handle = e.id;
    return AMDMSM_OK;
}

// AMDMSM_BASE_CACHE_MB=<cap>: host-buffer calls register the base vectors they see (keyed on
// pointer, length, stride and form) up to <cap> MiB of HBM, least recently used first out.
// Off by default: the caller must not modify a cached vector in place without
// amdmsm_invalidate_bases (the reference re-reads the bases on every call).
size_t auto_cache_cap_bytes() {
    static const size_t cap = getenv("AMDMSM_BASE_CACHE_MB") ? (size_t)atoll(getenv("AMDMSM_BASE_CACHE_MB")) << 20 : 0;
    return cap;
}

void drop_entry(amdmsm_ctx *ctx, size_t i) {
    if (ctx->bases[i].d_aff) (void)hipFree(ctx->bases[i].d_aff);
    if (ctx->bases[i].d_endo) (void)hipFree(ctx->bases[i].d_endo);
    ctx->bases.erase(ctx->bases.begin() + (long)i);
}

void *auto_cache_bases(amdmsm_ctx *ctx, const group_vtable *vt, const void *bases, size_t stride, int form, size_t n) {
    // (an entry is charged twice its affine bytes: the phi(P) records a split MSM attaches later count too)
    const size_t cap = auto_cache_cap_bytes(), need = 2 * n * (size_t)vt->el_words * 8;
    if (!cap || need > cap || n < 1024) return nullptr;
    (void)hipStreamSynchronize(ctx->stream);
    for (;;) {
        size_t used = 0, lru = (size_t)-1;
        for (size_t i = 0; i < ctx->bases.size(); ++i) {
            const base_entry &e = ctx->bases[i];
            if (!e.automatic) continue;
            used += 2 * e.n * e.aff_bytes;
            if (e.pinned) continue;   // a vector the running batch already resolved: its device copy must outlive the call
            if (lru == (size_t)-1 || e.last_use < ctx->bases[lru].last_use) lru = i;
        }
        if (used + need <= cap) break;
        if (lru == (size_t)-1) return nullptr;   // what is left is pinned by the running call: no room, the caller uploads
        drop_entry(ctx, lru);
    }
    if (register_bases_impl(ctx, vt, bases, stride, form, n, true, nullptr) != AMDMSM_OK) return nullptr;
    return ctx->bases.back().d_aff;
}

struct bases_upload {
    amdmsm_ctx *ctx;
    const group_vtable *vt;
    const void *bases;
    size_t stride, n;
    int form;
};
// runs once the sort is enqueued: bases H2D + import on the copy stream (msm_hook)
int upload_bases_hook(void *arg, hipEvent_t *wait_for) {
    bases_upload &u = *(bases_upload *)arg;
    amdmsm_ctx *ctx = u.ctx;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->hb_src.p, u.bases, u.n * u.stride, hipMemcpyHostToDevice, ctx->copy_stream));
    u.vt->import_bases(ctx->copy_stream, (const uint32_t *)ctx->hb_src.p, u.stride / 4, u.form == AMDMSM_FORM_SPECIAL, u.n,
                       (uint32_t *)ctx->hb_aff.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->bases_ready, ctx->copy_stream));
    *wait_for = ctx->bases_ready;
    return AMDMSM_OK;
}

// One MSM over host vectors, enqueued on the context stream; the result is left in
// ctx->hb_out (and the scalar statistics in ctx->hb_stats) -- the caller copies it back.
// Device buffers are the context's grow-only staging buffers: no allocation in steady state.
int host_msm_enqueue(amdmsm_ctx *ctx, const group_vtable *vt, const void *bases_xyz, size_t stride, int base_form,
                     const void *scalars, size_t n, const amdmsm_opts *opts, bool want_stats, bool clear_stats = true) {
    const size_t xyz_bytes = (size_t)vt->el_words * 12, aff_bytes = (size_t)vt->el_words * 8;
    const size_t fr_bytes = (size_t)vt->fr_words * 4;
    hipStream_t st = ctx->stream;
    int rc = ensure_buf(ctx, ctx->hb_out, xyz_bytes);
    if (rc) return rc;
    if (want_stats) {
        rc = ensure_buf(ctx, ctx->hb_stats, 16);
        if (rc) return rc;
        if (clear_stats) HIP_TRY(ctx, hipMemsetAsync(ctx->hb_stats.p, 0, 16, st));
    }
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    else o.out_form = AMDMSM_OUT_LIBFF;
    o.stream = st;
    void *d_aff = nullptr;
    const uint32_t *d_endo = nullptr;
    bases_upload up{ctx, vt, bases_xyz, stride, n, base_form};
    msm_hook hook;
    if (n) {
        rc = ensure_buf(ctx, ctx->hb_sc, n * fr_bytes);
        if (rc) return rc;
        base_entry *be = nullptr;
        size_t first = 0;
        d_aff = find_resident_bases(ctx, vt, bases_xyz, stride, base_form, n, &be, &first);
        if (!d_aff && auto_cache_bases(ctx, vt, bases_xyz, stride, base_form, n))
            d_aff = find_resident_bases(ctx, vt, bases_xyz, stride, base_form, n, &be, &first);
        if (d_aff && be && use_endomorphism(vt, n, &o, 0)) {
            // resident bases keep their phi(P) records too: built once (whole vector), then every
            // split MSM over them skips k_endo_points
            if (!be->d_endo) {
                if (hipMalloc(&be->d_endo, std::max<size_t>(be->n * aff_bytes, 256)) == hipSuccess) {
                    vt->endo_points(st, (const uint32_t *)be->d_aff, be->n, (uint32_t *)be->d_endo);
                    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
                        (void)hipFree(be->d_endo);
                        be->d_endo = nullptr;
                    }
                } else {
                    be->d_endo = nullptr;   // no room: the per-call kernel does it
                    (void)hipGetLastError();
                }
            }
            if (be->d_endo) d_endo = (const uint32_t *)((const char *)be->d_endo + first * aff_bytes);
        }
        if (!d_aff) {
            rc = ensure_buf(ctx, ctx->hb_src, n * stride);
            if (rc == AMDMSM_OK) rc = ensure_buf(ctx, ctx->hb_aff, n * aff_bytes);
            if (rc) return rc;
            d_aff = ctx->hb_aff.p;
            hook.fn = upload_bases_hook;
            hook.arg = &up;
        }
        HIP_TRY(ctx, hipMemcpyAsync(ctx->hb_sc.p, scalars, n * fr_bytes, hipMemcpyHostToDevice, st));
        if (want_stats) {
            vt->scalar_stats(st, (const uint32_t *)ctx->hb_sc.p, n, o.scalars_plain ? 0 : 1, (uint32_t *)ctx->hb_stats.p);
            HIP_TRY(ctx, hipGetLastError());
        }
    }
    return msm_device_impl(ctx, vt, (const uint32_t *)d_aff, (const uint32_t *)ctx->hb_sc.p, n, (uint32_t *)ctx->hb_out.p, &o,
                           0, &hook, d_endo);
}

// one partial point from the device that produced it to the combining device (xGMI peer copy)
// (AMDMSM_FORCE_PEER_COPY=1: the peer-copy call also between two contexts of ONE device, so that a
// one-GPU test box executes the exchange branch of the multi-device entries)
hipError_t copy_partial(hipStream_t st, void *dst, int dst_dev, const void *src, int src_dev, size_t bytes) {
    static const bool force_peer = getenv("AMDMSM_FORCE_PEER_COPY") && atoi(getenv("AMDMSM_FORCE_PEER_COPY")) != 0;
    if (dst_dev == src_dev && !force_peer) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}

// host_msm_enqueue for inputs of any length: above max_range_points() the reference's chunk loop
// (multiexp.tcc:655-687), range after range through the same staging buffers (each range is
// finished before the next one reuses them), the partial points summed at the end.  The result is
// left in ctx->hb_out in the requested form, the 0 / 1 counters of all ranges in ctx->hb_stats.
int host_msm_ranges(amdmsm_ctx *ctx, const group_vtable *vt, const void *bases_xyz, size_t stride, int base_form,
                    const void *scalars, size_t n, const amdmsm_opts *opts, bool want_stats) {
    const size_t maxr = max_range_points();
    if (n <= maxr) return host_msm_enqueue(ctx, vt, bases_xyz, stride, base_form, scalars, n, opts, want_stats);
    const size_t parts = (n + maxr - 1) / maxr;
    if (parts > MAX_RANGES) return fail(ctx, AMDMSM_ERR_TOO_LARGE, "input too large for one call");
    const size_t one = (n + parts - 1) / parts;   // <= maxr (see msm_device_ranges)
    const size_t xyz_bytes = (size_t)vt->el_words * 12, fr_bytes = (size_t)vt->fr_words * 4;
    int rc = ensure_partials(ctx);
    if (rc) return rc;
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    else o.out_form = AMDMSM_OUT_LIBFF;
    const int form = o.out_form;
    o.out_form = AMDMSM_OUT_JACOBIAN;
    hipStream_t st = ctx->stream;
    for (size_t k = 0; k < parts; ++k) {
        const size_t lo = k * one, cnt = (k == parts - 1) ? n - lo : one;
        rc = host_msm_enqueue(ctx, vt, (const char *)bases_xyz + lo * stride, stride, base_form,
                              (const char *)scalars + lo * fr_bytes, cnt, &o, want_stats, k == 0);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemcpyAsync((char *)ctx->chunk_partials + k * xyz_bytes, ctx->hb_out.p, xyz_bytes,
                                    hipMemcpyDeviceToDevice, st));
        HIP_TRY(ctx, hipStreamSynchronize(st));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    }
    vt->sum_points(st, (const uint32_t *)ctx->chunk_partials, (int)parts, form, (uint32_t *)ctx->hb_out.p);
    HIP_TRY(ctx, hipGetLastError());
    return AMDMSM_OK;
}

int check_host_args(amdmsm_ctx *ctx, const group_vtable *vt, const void *bases_xyz, size_t &stride, const void *scalars,
                    size_t n, const void *out_xyz) {
    if (!out_xyz || (n && (!bases_xyz || !scalars))) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
    const size_t xyz_bytes = (size_t)vt->el_words * 12;
    if (stride == 0) stride = xyz_bytes;
    if (stride % 16 || stride < xyz_bytes) return fail(ctx, AMDMSM_ERR_BAD_ARG, "base stride");
    return AMDMSM_OK;
}

int host_multi_exp(amdmsm_ctx *ctx, const group_vtable *vt, const void *bases_xyz, size_t stride, int base_form,
                   const void *scalars, size_t n, void *out_xyz, const amdmsm_opts *opts, size_t stats[3]) {
    int rc = check_host_args(ctx, vt, bases_xyz, stride, scalars, n, out_xyz);
    if (rc) return rc;
    rc = host_msm_ranges(ctx, vt, bases_xyz, stride, base_form, scalars, n, opts, stats != nullptr);
    if (rc) {
        (void)hipDeviceSynchronize();   // nothing of this call may still touch the staging buffers
        return rc;
    }
    hipStream_t st = ctx->stream;
    uint32_t hs[4] = {};
    HIP_TRY(ctx, hipMemcpyAsync(out_xyz, ctx->hb_out.p, (size_t)vt->el_words * 12, hipMemcpyDeviceToHost, st));
    if (stats) HIP_TRY(ctx, hipMemcpyAsync(hs, ctx->hb_stats.p, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (stats) {
        stats[0] = hs[0];
        stats[1] = hs[1];
        stats[2] = n - hs[0] - hs[1];
    }
    return AMDMSM_OK;
}

}   // namespace

extern "C" {

int amdmsm_multi_exp(amdmsm_ctx *ctx, int curve, int group, const void *bases_xyz, size_t base_stride_bytes,
                     int base_form, const void *scalars, size_t n, void *out_xyz, const amdmsm_opts *opts) {
    GET_VT(ctx, curve, group);
    CHECK_OPTS(ctx, opts);
    return host_multi_exp(ctx, vt, bases_xyz, base_stride_bytes, base_form, scalars, n, out_xyz, opts, nullptr);
}

int amdmsm_multi_exp_filter_one_zero(amdmsm_ctx *ctx, int curve, int group, const void *bases_xyz,
                                     size_t base_stride_bytes, int base_form, const void *scalars, size_t n,
                                     void *out_xyz, const amdmsm_opts *opts, size_t stats[3]) {
    // multiexp.tcc:690-757 separates scalars equal to 0 and 1 before the Pippenger
    // pass.  On the device a zero scalar recodes to all-zero digits (no work) and a
    // one lands in bucket 1 of window 0, so the same kernels compute the same sum;
    // only the three statistics the reference prints need the classification, which
    // k_scalar_stats counts on the device from the scalars already in HBM.
    GET_VT(ctx, curve, group);
    CHECK_OPTS(ctx, opts);
    return host_multi_exp(ctx, vt, bases_xyz, base_stride_bytes, base_form, scalars, n, out_xyz, opts, stats);
}

// k multi_exp calls of one group and length as ONE batch (amdmsm_msm_device_batch over host vectors): scalars over PCIe
// per MSM, bases from the resident copies where registered (amdmsm_register_bases) and uploaded + imported otherwise.
int amdmsm_multi_exp_batch(amdmsm_ctx *ctx, int curve, int group, int k, const void *const *bases_xyz,
                           size_t base_stride_bytes, int base_form, const void *const *scalars, size_t n,
                           void *const *out_xyz, const amdmsm_opts *opts) {
    GET_VT(ctx, curve, group);
    CHECK_OPTS(ctx, opts);
    if (k < 1 || k > MAX_BATCH || !bases_xyz || !scalars || !out_xyz) return fail(ctx, AMDMSM_ERR_BAD_ARG, "batch of 1 .. 8 MSMs");
    size_t stride = base_stride_bytes;
    for (int j = 0; j < k; ++j) {
        const int rcj = check_host_args(ctx, vt, bases_xyz[j], stride, scalars[j], n, out_xyz[j]);
        if (rcj) return rcj;
    }
    if (k == 1 || n == 0 || n > max_range_points() || (size_t)k * n >= ((size_t)1 << 30)) {
        for (int j = 0; j < k; ++j) {
            const int rcj = host_multi_exp(ctx, vt, bases_xyz[j], stride, base_form, scalars[j], n, out_xyz[j], opts, nullptr);
            if (rcj) return rcj;
        }
        return AMDMSM_OK;
    }
    const size_t xyz_bytes = (size_t)vt->el_words * 12, aff_bytes = (size_t)vt->el_words * 8, fr_bytes = (size_t)vt->fr_words * 4;
    hipStream_t st = ctx->stream;
    int rc = ensure_buf(ctx, ctx->hb_out, (size_t)k * xyz_bytes);
    if (rc == AMDMSM_OK) rc = ensure_buf(ctx, ctx->hb_sc, (size_t)k * n * fr_bytes);
    if (rc) return rc;
    const uint32_t *d_b[MAX_BATCH], *d_s[MAX_BATCH];
    uint32_t *d_o[MAX_BATCH];
    int missing = 0;
    // Every resident vector this batch resolves is pinned until the call returns: registering vector j under
    // AMDMSM_BASE_CACHE_MB evicts least-recently-used automatic entries (hipFree), which must never be the copy behind
    // d_b[j'] of an earlier j'.  A vector that does not fit beside the pinned ones is uploaded like an unregistered one.
    struct unpin_all {
        amdmsm_ctx *c;
        ~unpin_all() {
            for (auto &e : c->bases) e.pinned = false;
        }
    } unpin{ctx};
    for (int j = 0; j < k; ++j) {
        base_entry *be = nullptr;
        d_b[j] = (const uint32_t *)find_resident_bases(ctx, vt, bases_xyz[j], stride, base_form, n, &be);
        if (!d_b[j] && auto_cache_bases(ctx, vt, bases_xyz[j], stride, base_form, n))
            d_b[j] = (const uint32_t *)find_resident_bases(ctx, vt, bases_xyz[j], stride, base_form, n, &be);
        if (d_b[j] && be) be->pinned = true;
        if (!d_b[j]) ++missing;
    }
    if (missing) {
        rc = ensure_buf(ctx, ctx->hb_src, n * stride);
        if (rc == AMDMSM_OK) rc = ensure_buf(ctx, ctx->hb_aff, (size_t)missing * n * aff_bytes);
        if (rc) return rc;
    }
    int slot = 0;
    for (int j = 0; j < k; ++j) {
        d_s[j] = (const uint32_t *)((char *)ctx->hb_sc.p + (size_t)j * n * fr_bytes);
        d_o[j] = (uint32_t *)((char *)ctx->hb_out.p + (size_t)j * xyz_bytes);
        HIP_TRY(ctx, hipMemcpyAsync((void *)d_s[j], scalars[j], n * fr_bytes, hipMemcpyHostToDevice, st));
        if (!d_b[j]) {
            uint32_t *aff = (uint32_t *)((char *)ctx->hb_aff.p + (size_t)slot++ * n * aff_bytes);
            HIP_TRY(ctx, hipMemcpyAsync(ctx->hb_src.p, bases_xyz[j], n * stride, hipMemcpyHostToDevice, st));
            vt->import_bases(st, (const uint32_t *)ctx->hb_src.p, stride / 4, base_form == AMDMSM_FORM_SPECIAL, n, aff);
            HIP_TRY(ctx, hipGetLastError());
            d_b[j] = aff;
        }
    }
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    else o.out_form = AMDMSM_OUT_LIBFF;
    o.stream = st;
    rc = msm_device_batch_impl(ctx, vt, k, d_b, d_s, n, d_o, &o);
    if (rc) {
        (void)hipDeviceSynchronize();
        return rc;
    }
    for (int j = 0; j < k; ++j) HIP_TRY(ctx, hipMemcpyAsync(out_xyz[j], d_o[j], xyz_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return AMDMSM_OK;
}

int amdmsm_register_bases(amdmsm_ctx *ctx, int curve, int group, const void *bases_xyz, size_t base_stride_bytes,
                          int base_form, size_t n, uint64_t *handle) {
    GET_VT(ctx, curve, group);
    if (!bases_xyz || !n) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null / empty base vector");
    const size_t xyz_bytes = (size_t)vt->el_words * 12;
    if (base_stride_bytes == 0) base_stride_bytes = xyz_bytes;
    if (base_stride_bytes % 16 || base_stride_bytes < xyz_bytes) return fail(ctx, AMDMSM_ERR_BAD_ARG, "base stride");
    // a re-registration of the same range replaces the old copy
    for (size_t i = ctx->bases.size(); i-- > 0;) {
        const base_entry &e = ctx->bases[i];
        if (e.host == (const char *)bases_xyz && e.curve == curve && e.group == group) drop_entry(ctx, i);
    }
    return register_bases_impl(ctx, vt, bases_xyz, base_stride_bytes, base_form, n, false, handle);
}

int amdmsm_unregister_bases(amdmsm_ctx *ctx, uint64_t handle) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    dev_guard g(ctx->device);
    for (size_t i = 0; i < ctx->bases.size(); ++i) {
        if (ctx->bases[i].id == handle) {
            HIP_TRY(ctx, hipDeviceSynchronize());
            drop_entry(ctx, i);
            return AMDMSM_OK;
        }
    }
    return fail(ctx, AMDMSM_ERR_BAD_ARG, "unknown base handle");
}

int amdmsm_invalidate_bases(amdmsm_ctx *ctx, const void *host_ptr, size_t bytes) {
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    dev_guard g(ctx->device);
    HIP_TRY(ctx, hipDeviceSynchronize());
    const char *lo = (const char *)host_ptr, *hi = lo + bytes;
    for (size_t i = ctx->bases.size(); i-- > 0;) {
        const base_entry &e = ctx->bases[i];
        if (!host_ptr || (lo < e.host + e.n * e.stride && hi > e.host)) drop_entry(ctx, i);
    }
    return AMDMSM_OK;
}

// multi_exp across several devices of one node (multiexp.tcc:655-687 with chunk = device):
// context k takes the contiguous range [k*one, (k+1)*one) (the last one the remainder), every
// range runs the whole single-GPU pipeline on its own device from its own host thread, the
// partial points are brought to the first device (hipMemcpyPeerAsync over xGMI) and summed there.
int amdmsm_multi_exp_filter_one_zero_multi(amdmsm_ctx *const *ctxs, int ndev, int curve, int group, const void *bases_xyz,
                                           size_t base_stride_bytes, int base_form, const void *scalars, size_t n,
                                           void *out_xyz, const amdmsm_opts *opts, size_t stats[3]) {
    if (!ctxs || ndev < 1 || ndev > 64) return AMDMSM_ERR_BAD_ARG;
    for (int k = 0; k < ndev; ++k) {
        if (!ctxs[k]) return AMDMSM_ERR_BAD_ARG;
        for (int j = 0; j < k; ++j) {
            if (ctxs[j] == ctxs[k]) return fail(ctxs[0], AMDMSM_ERR_BAD_ARG, "the same context twice");
        }
    }
    amdmsm_ctx *c0 = ctxs[0];
    CHECK_OPTS(c0, opts);
    const group_vtable *vt = find_vt(curve, group);
    if (!vt) return fail(c0, AMDMSM_ERR_UNSUPPORTED, "unknown curve/group");
    if (ndev == 1 || n < (size_t)ndev) {
        if (stats) return amdmsm_multi_exp_filter_one_zero(c0, curve, group, bases_xyz, base_stride_bytes, base_form, scalars,
                                                           n, out_xyz, opts, stats);
        return amdmsm_multi_exp(c0, curve, group, bases_xyz, base_stride_bytes, base_form, scalars, n, out_xyz, opts);
    }
    // lock order = argument order; every caller passing the same list is deadlock-free
    std::vector<std::unique_lock<std::recursive_mutex>> locks;
    for (int k = 0; k < ndev; ++k) locks.emplace_back(ctxs[k]->mu);
    size_t stride = base_stride_bytes;
    int rc = check_host_args(c0, vt, bases_xyz, stride, scalars, n, out_xyz);
    if (rc) return rc;
    const size_t xyz_bytes = (size_t)vt->el_words * 12, fr_bytes = (size_t)vt->fr_words * 4;
    const size_t one = n / (size_t)ndev;
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    else o.out_form = AMDMSM_OUT_LIBFF;
    const int final_form = o.out_form;
    o.out_form = AMDMSM_OUT_JACOBIAN;
    std::vector<int> rcs((size_t)ndev, AMDMSM_OK);
    auto work = [&](int k) {
        amdmsm_ctx *ctx = ctxs[k];
        dev_guard g(ctx->device);
        const size_t lo = (size_t)k * one, cnt = (k == ndev - 1) ? n - lo : one;
        // every device classifies the scalars of its own range (multiexp.tcc:713-733); the counts are added up below
        int r = host_msm_ranges(ctx, vt, (const char *)bases_xyz + lo * stride, stride, base_form,
                                (const char *)scalars + lo * fr_bytes, cnt, &o, stats != nullptr);
        if (r == AMDMSM_OK && hipEventRecord(ctx->host_done, ctx->stream) != hipSuccess) r = AMDMSM_ERR_HIP;
        rcs[(size_t)k] = r;
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < ndev; ++k) threads.emplace_back(work, k);
    work(0);
    for (auto &t : threads) t.join();
    int failed = -1;
    for (int k = 0; k < ndev; ++k) {
        if (rcs[(size_t)k] && failed < 0) failed = k;
    }
    dev_guard g0(c0->device);
    if (failed >= 0) {
        for (int k = 0; k < ndev; ++k) {
            dev_guard g(ctxs[k]->device);
            (void)hipDeviceSynchronize();
        }
        // amdmsm_last_error is asked of the first context: carry the failing range's message over
        return fail(c0, rcs[(size_t)failed], "range " + std::to_string(failed) + ": " + ctxs[failed]->err);
    }
    rc = ensure_partials(c0);
    if (rc) return rc;
    hipStream_t st = c0->stream;
    for (int k = 0; k < ndev; ++k) {
        HIP_TRY(c0, hipStreamWaitEvent(st, ctxs[k]->host_done, 0));
        HIP_TRY(c0, copy_partial(st, (char *)c0->chunk_partials + (size_t)k * xyz_bytes, c0->device, ctxs[k]->hb_out.p,
                                 ctxs[k]->device, xyz_bytes));
    }
    // (device 0's own partial has been copied out of hb_out by the loop above, in stream order)
    vt->sum_points(st, (const uint32_t *)c0->chunk_partials, ndev, final_form, (uint32_t *)c0->hb_out.p);
    HIP_TRY(c0, hipGetLastError());
    HIP_TRY(c0, hipMemcpyAsync(out_xyz, c0->hb_out.p, xyz_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(c0, hipStreamSynchronize(st));
    size_t zeros = 0, ones = 0;
    for (int k = 0; k < ndev; ++k) {
        dev_guard g(ctxs[k]->device);
        HIP_TRY(c0, hipStreamSynchronize(ctxs[k]->copy_stream));
        HIP_TRY(c0, hipStreamSynchronize(ctxs[k]->stream));
        if (stats) {
            uint32_t hs[4] = {};
            HIP_TRY(c0, hipMemcpy(hs, ctxs[k]->hb_stats.p, 16, hipMemcpyDeviceToHost));
            zeros += hs[0];
            ones += hs[1];
        }
    }
    if (stats) {
        stats[0] = zeros;
        stats[1] = ones;
        stats[2] = n - zeros - ones;
    }
    return AMDMSM_OK;
}

int amdmsm_multi_exp_multi(amdmsm_ctx *const *ctxs, int ndev, int curve, int group, const void *bases_xyz,
                           size_t base_stride_bytes, int base_form, const void *scalars, size_t n, void *out_xyz,
                           const amdmsm_opts *opts) {
    return amdmsm_multi_exp_filter_one_zero_multi(ctxs, ndev, curve, group, bases_xyz, base_stride_bytes, base_form, scalars, n,
                                                  out_xyz, opts, nullptr);
}

// Device-resident counterpart: context k holds its own range (compact affine bases + scalars)
// in its own HBM; partial k is reduced on device k and the partials are summed on device 0.
int amdmsm_msm_device_multi(amdmsm_ctx *const *ctxs, int ndev, int curve, int group, const void *const *d_bases_affine,
                            const void *const *d_scalars, const size_t *counts, void *d_out_xyz_dev0,
                            const amdmsm_opts *opts) {
    if (!ctxs || ndev < 1 || ndev > 64 || !d_bases_affine || !d_scalars || !counts || !d_out_xyz_dev0) return AMDMSM_ERR_BAD_ARG;
    for (int k = 0; k < ndev; ++k) {
        if (!ctxs[k]) return AMDMSM_ERR_BAD_ARG;
        for (int j = 0; j < k; ++j) {
            if (ctxs[j] == ctxs[k]) return fail(ctxs[0], AMDMSM_ERR_BAD_ARG, "the same context twice");
        }
    }
    amdmsm_ctx *c0 = ctxs[0];
    CHECK_OPTS(c0, opts);
    const group_vtable *vt = find_vt(curve, group);
    if (!vt) return fail(c0, AMDMSM_ERR_UNSUPPORTED, "unknown curve/group");
    std::vector<std::unique_lock<std::recursive_mutex>> locks;
    for (int k = 0; k < ndev; ++k) locks.emplace_back(ctxs[k]->mu);
    const size_t xyz_bytes = (size_t)vt->el_words * 12;
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    else o.out_form = AMDMSM_OUT_LIBFF;
    const int final_form = o.out_form;
    hipStream_t user_stream = (hipStream_t)o.stream;
    o.out_form = AMDMSM_OUT_JACOBIAN;
    o.stream = nullptr;
    // The ranges run on the contexts' own streams.  Inputs the caller produced on opts->stream are
    // ordered before them: every context stream waits for an event recorded on that stream now.
    if (user_stream) {
        dev_guard g(c0->device);
        HIP_TRY(c0, hipEventRecord(c0->bases_ready, user_stream));
    }
    // kernels are asynchronous: one host thread enqueues all devices
    int rc = AMDMSM_OK, failed = -1;
    for (int k = 0; k < ndev && rc == AMDMSM_OK; ++k) {
        amdmsm_ctx *ctx = ctxs[k];
        dev_guard g(ctx->device);
        failed = k;
        rc = ensure_buf(ctx, ctx->hb_out, xyz_bytes);
        if (rc) break;
        if (counts[k] && (!d_bases_affine[k] || !d_scalars[k])) {
            rc = fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
            break;
        }
        if (user_stream && hipStreamWaitEvent(ctx->stream, c0->bases_ready, 0) != hipSuccess) {
            rc = fail(ctx, AMDMSM_ERR_HIP, "hipStreamWaitEvent");
            break;
        }
        rc = msm_device_ranges(ctx, vt, (const uint32_t *)d_bases_affine[k], (const uint32_t *)d_scalars[k], counts[k],
                               (uint32_t *)ctx->hb_out.p, &o);
        if (rc) break;
        if (hipEventRecord(ctx->host_done, ctx->stream) != hipSuccess) rc = fail(ctx, AMDMSM_ERR_HIP, "hipEventRecord");
    }
    if (rc) {
        // nothing of this call may still be in flight when the caller reads the error
        for (int k = 0; k < ndev; ++k) {
            dev_guard g(ctxs[k]->device);
            (void)hipDeviceSynchronize();
        }
        return fail(c0, rc, "range " + std::to_string(failed) + ": " + ctxs[failed]->err);
    }
    dev_guard g0(c0->device);
    rc = ensure_partials(c0);
    if (rc) return rc;
    hipStream_t st = user_stream ? user_stream : c0->stream;
    for (int k = 0; k < ndev; ++k) {
        HIP_TRY(c0, hipStreamWaitEvent(st, ctxs[k]->host_done, 0));
        HIP_TRY(c0, copy_partial(st, (char *)c0->chunk_partials + (size_t)k * xyz_bytes, c0->device, ctxs[k]->hb_out.p,
                                 ctxs[k]->device, xyz_bytes));
    }
    vt->sum_points(st, (const uint32_t *)c0->chunk_partials, ndev, final_form, (uint32_t *)d_out_xyz_dev0);
    HIP_TRY(c0, hipGetLastError());
    HIP_TRY(c0, hipStreamSynchronize(st));   // the partials buffer is shared by the context
    return AMDMSM_OK;
}

}   // extern "C"

namespace {
// multi_exp_stream (multiexp_stream.hpp:25-33, multiexp_stream.tcc:164-191): the bases arrive
// through a reader in libff's on-disk format and never have to be resident at once.  Chunks of
// `chunk_points` records are read into pinned staging buffers, copied and decoded on the
// device, and reduced to one partial point each on alternating streams / workspace slots, so
// reading chunk k+1 overlaps the MSM of chunk k; the partials are summed at the end
// (multiexp.tcc:681-687).
// recs = records per scalar in the stream: 1 for multi_exp_stream; for the precompute variant the
// num_digits multiples [2^(jc)]P of each base, consumed with window size precompute_c
// (element_buffers_from_stream_producer, multiexp_stream.tcc:33-49).
// compressed: the records hold X and two flag bits only (compression_on, curve_serialization.tcc:
// 103-166); Y is recovered on the device by a square root.
int stream_impl(amdmsm_ctx *ctx, int curve, int group, amdmsm_read_fn read, void *read_ctx, const void *scalars,
                size_t n, size_t chunk_points, void *out_xyz, const amdmsm_opts *opts, size_t recs,
                size_t precompute_c, bool compressed = false) {
    if (!ctx || !read || !out_xyz || (n && !scalars)) return AMDMSM_ERR_BAD_ARG;
    CHECK_OPTS(ctx, opts);
    const group_vtable *vt = find_vt(curve, group);
    if (!vt) return fail(ctx, AMDMSM_ERR_UNSUPPORTED, "unknown curve/group");
    const size_t xyz_bytes = (size_t)vt->el_words * 12, aff_bytes = (size_t)vt->el_words * 8 * recs;
    const size_t rec_bytes = compressed ? aff_bytes / 2 : aff_bytes;   // bytes in the stream per scalar
    const size_t fr_bytes = (size_t)vt->fr_words * 4;
    if (chunk_points == 0) chunk_points = std::max<size_t>(((size_t)1 << 20) / recs, 1024);
    if (chunk_points > n && n) chunk_points = n;
    const size_t nchunks = n ? (n + chunk_points - 1) / chunk_points : 0;
    constexpr int NB = stream_state::NB;
    // the call owns the context from here on: it changes the pipeline depth and the slot rotation
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    const int old_depth = ctx->depth;
    dev_guard guard(ctx->device);
    stream_state &ss = ctx->ss;
    // Staging lives in the context and only grows: pinned host buffers, device buffers and the two streams are
    // set up by the first call (2 x 64 MiB of pinned memory cost ~40 ms to allocate), later calls reuse them.
    auto finish = [&](int rc_) {
        (void)hipDeviceSynchronize();
        (void)amdmsm_set_pipeline_depth(ctx, old_depth);
        return rc_;
    };
#define TRY_S(expr)                                                                                   \
    do {                                                                                              \
        hipError_t e_ = (expr);                                                                       \
        if (e_ != hipSuccess) return finish(fail(ctx, AMDMSM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); \
    } while (0)
    int rc = amdmsm_set_pipeline_depth(ctx, NB);
    if (rc) return rc;
    rc = ensure_buf(ctx, ss.partials, (nchunks + 1) * xyz_bytes);
    if (rc == AMDMSM_OK) rc = ensure_buf(ctx, ss.status, 16);
    for (int b = 0; b < NB && nchunks && rc == AMDMSM_OK; ++b) {
        if (ss.h_bytes[b] < chunk_points * aff_bytes) {
            (void)hipDeviceSynchronize();
            if (ss.h_stage[b]) (void)hipHostFree(ss.h_stage[b]);
            ss.h_stage[b] = nullptr;
            ss.h_bytes[b] = 0;
            TRY_S(hipHostMalloc(&ss.h_stage[b], chunk_points * aff_bytes, hipHostMallocDefault));
            ss.h_bytes[b] = chunk_points * aff_bytes;
        }
        rc = ensure_buf(ctx, ss.d_raw[b], chunk_points * aff_bytes);
        if (rc == AMDMSM_OK) rc = ensure_buf(ctx, ss.d_aff[b], chunk_points * aff_bytes);
        if (rc == AMDMSM_OK) rc = ensure_buf(ctx, ss.d_sc[b], chunk_points * fr_bytes);
        if (rc == AMDMSM_OK && !ss.streams[b]) TRY_S(hipStreamCreateWithFlags(&ss.streams[b], hipStreamNonBlocking));
    }
    if (rc) return finish(rc);
    void *d_partials = ss.partials.p, *d_status = ss.status.p;
    TRY_S(hipMemsetAsync(d_status, 0, 16, ctx->stream));
    TRY_S(hipStreamSynchronize(ctx->stream));
    amdmsm_opts o = AMDMSM_OPTS_INIT;
    if (opts) o = *opts;
    const int final_form = opts ? opts->out_form : AMDMSM_OUT_LIBFF;
    o.out_form = AMDMSM_OUT_JACOBIAN;
    for (size_t k = 0; k < nchunks; ++k) {
        const int b = (int)(k % NB);
        const size_t lo = k * chunk_points, cnt = std::min(chunk_points, n - lo);
        TRY_S(hipStreamSynchronize(ss.streams[b]));   // staging buffer b is free again
        const size_t want = cnt * rec_bytes;
        size_t got = 0;
        while (got < want) {
            const size_t r = read(read_ctx, (char *)ss.h_stage[b] + got, want - got);
            if (r == 0) break;
            got += r;
        }
        if (got != want) return finish(fail(ctx, AMDMSM_ERR_BAD_ARG, "base-element stream ended early"));
        TRY_S(hipMemcpyAsync(ss.d_raw[b].p, ss.h_stage[b], want, hipMemcpyHostToDevice, ss.streams[b]));
        TRY_S(hipMemcpyAsync(ss.d_sc[b].p, (const char *)scalars + lo * fr_bytes, cnt * fr_bytes, hipMemcpyHostToDevice,
                             ss.streams[b]));
        if (compressed)
            vt->disk_decode_compressed(ss.streams[b], (const uint32_t *)ss.d_raw[b].p, cnt * recs, (uint32_t *)ss.d_aff[b].p,
                                       (uint32_t *)d_status);
        else
            vt->disk_decode(ss.streams[b], (const uint32_t *)ss.d_raw[b].p, cnt * recs, (uint32_t *)ss.d_aff[b].p);
        o.stream = ss.streams[b];
        if (precompute_c)
            rc = amdmsm_msm_precomputed_device(ctx, curve, group, ss.d_aff[b].p, ss.d_sc[b].p, cnt, precompute_c, recs,
                                               (char *)d_partials + k * xyz_bytes, &o);
        else
            rc = amdmsm_msm_device(ctx, curve, group, ss.d_aff[b].p, ss.d_sc[b].p, cnt, (char *)d_partials + k * xyz_bytes, &o);
        if (rc) return finish(rc);
    }
    TRY_S(hipDeviceSynchronize());
    vt->sum_points(ctx->stream, (const uint32_t *)d_partials, (int)nchunks, final_form,
                   (uint32_t *)((char *)d_partials + nchunks * xyz_bytes));
    unsigned status = 0;
    TRY_S(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, ctx->stream));
    TRY_S(hipStreamSynchronize(ctx->stream));
    // some X is not the abscissa of a curve point: no result (the reference's sqrt would not return)
    if (status) return finish(fail(ctx, AMDMSM_ERR_BAD_ARG, "compressed base element is not on the curve"));
    TRY_S(hipMemcpyAsync(out_xyz, (char *)d_partials + nchunks * xyz_bytes, xyz_bytes, hipMemcpyDeviceToHost, ctx->stream));
    TRY_S(hipStreamSynchronize(ctx->stream));
    return finish(AMDMSM_OK);
#undef TRY_S
}

// file reader of the *_file entries: the chunk is read by several threads at once, each with pread on its own
// slice (one thread copies from the page cache at ~3.5 GB/s, less than the PCIe link takes)
struct file_src {
    int fd = -1;
    size_t pos = 0;
};
size_t file_reader(void *ctxp, void *dst, size_t bytes) {
    file_src &f = *(file_src *)ctxp;
    constexpr size_t SLICE = (size_t)8 << 20;
    static const unsigned max_threads = [] {
        const char *e = getenv("AMDMSM_READ_THREADS");
        const unsigned hw = std::thread::hardware_concurrency();
        const unsigned def = std::min(8u, std::max(1u, hw / 2));
        return e && atoi(e) > 0 ? (unsigned)atoi(e) : def;
    }();
    const size_t slices = (bytes + SLICE - 1) / SLICE;
    const unsigned nt = (unsigned)std::min<size_t>(max_threads, slices);
    std::vector<size_t> got((size_t)std::max(nt, 1u), 0);
    auto work = [&](unsigned t) {
        for (size_t sidx = t; sidx < slices; sidx += nt) {
            size_t off = sidx * SLICE;
            const size_t end = std::min(bytes, off + SLICE);
            while (off < end) {
                const ssize_t r = pread(f.fd, (char *)dst + off, end - off, (off_t)(f.pos + off));
                if (r <= 0) return;
                off += (size_t)r;
                got[t] += (size_t)r;
            }
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    size_t total = 0;
    for (size_t g : got) total += g;
    if (total != bytes) {
        // short file: report the contiguous prefix only (the caller treats a short read as the end of the stream)
        struct stat st;
        const size_t size = fstat(f.fd, &st) == 0 ? (size_t)st.st_size : 0;
        total = size > f.pos ? std::min(bytes, size - f.pos) : 0;
        if (total == bytes) total = 0;
    }
    f.pos += total;
    return total;
}

int stream_file_impl(amdmsm_ctx *ctx, int curve, int group, const char *path, size_t offset_bytes, const void *scalars,
                     size_t n, size_t chunk_points, void *out_xyz, const amdmsm_opts *opts, size_t recs,
                     size_t precompute_c, bool compressed = false) {
    if (!ctx || !path) return AMDMSM_ERR_BAD_ARG;
    file_src f;
    f.fd = open(path, O_RDONLY);
    if (f.fd < 0) return fail(ctx, AMDMSM_ERR_BAD_ARG, std::string("cannot open ") + path);
    f.pos = offset_bytes;
    const int rc = stream_impl(ctx, curve, group, file_reader, &f, scalars, n, chunk_points, out_xyz, opts, recs,
                               precompute_c, compressed);
    close(f.fd);
    return rc;
}
}  // namespace

extern "C" {

int amdmsm_multi_exp_stream(amdmsm_ctx *ctx, int curve, int group, amdmsm_read_fn read, void *read_ctx,
                            const void *scalars, size_t n, size_t chunk_points, void *out_xyz,
                            const amdmsm_opts *opts) {
    return stream_impl(ctx, curve, group, read, read_ctx, scalars, n, chunk_points, out_xyz, opts, 1, 0);
}

int amdmsm_multi_exp_stream_file(amdmsm_ctx *ctx, int curve, int group, const char *path, size_t offset_bytes,
                                 const void *scalars, size_t n, size_t chunk_points, void *out_xyz,
                                 const amdmsm_opts *opts) {
    return stream_file_impl(ctx, curve, group, path, offset_bytes, scalars, n, chunk_points, out_xyz, opts, 1, 0);
}

// multi_exp_stream<form_montgomery, compression_on, G, Fr> (multiexp_stream.hpp:25-27 with
// Comp = compression_on; records of curve_serialization.tcc:103-133)
int amdmsm_multi_exp_stream_compressed(amdmsm_ctx *ctx, int curve, int group, amdmsm_read_fn read, void *read_ctx,
                                       const void *scalars, size_t n, size_t chunk_points, void *out_xyz,
                                       const amdmsm_opts *opts) {
    return stream_impl(ctx, curve, group, read, read_ctx, scalars, n, chunk_points, out_xyz, opts, 1, 0, true);
}

int amdmsm_multi_exp_stream_compressed_file(amdmsm_ctx *ctx, int curve, int group, const char *path, size_t offset_bytes,
                                            const void *scalars, size_t n, size_t chunk_points, void *out_xyz,
                                            const amdmsm_opts *opts) {
    return stream_file_impl(ctx, curve, group, path, offset_bytes, scalars, n, chunk_points, out_xyz, opts, 1, 0, true);
}

// group_read<encoding_binary, form_montgomery, Comp> over an array of records already in HBM
// (curve_serialization.tcc:78-101 / 134-166): n compact affine points out.
int amdmsm_disk_decode_device(amdmsm_ctx *ctx, int curve, int group, const void *d_records, size_t n, int compressed,
                              void *d_dst_affine, unsigned *status) {
    GET_VT(ctx, curve, group);
    if (n && (!d_records || !d_dst_affine)) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
    hipStream_t st = ctx->stream;
    unsigned hs = 0;
    if (compressed) {
        int rc = ensure_buf(ctx, ctx->hb_stats, 16);
        if (rc) return rc;
        HIP_TRY(ctx, hipMemsetAsync(ctx->hb_stats.p, 0, 16, st));
        vt->disk_decode_compressed(st, (const uint32_t *)d_records, n, (uint32_t *)d_dst_affine, (uint32_t *)ctx->hb_stats.p);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(&hs, ctx->hb_stats.p, 4, hipMemcpyDeviceToHost, st));
    } else {
        vt->disk_decode(st, (const uint32_t *)d_records, n, (uint32_t *)d_dst_affine);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (status) *status = hs;
    return AMDMSM_OK;
}

// multi_exp_stream_with_precompute (multiexp_stream.hpp:29-42, multiexp_stream.tcc:193-223): the
// stream holds, per base, the num_digits = ceil(Fr::num_bits / c) multiples [2^(jc)]P; every
// (scalar digit, multiple) pair lands in ONE bucket set and no doublings are needed.  Like the
// reference, a carry out of the last digit is dropped (it cannot occur when the top digit has
// spare bits, e.g. the 254-bit alt_bn128 Fr with c = 16).
int amdmsm_multi_exp_stream_with_precompute(amdmsm_ctx *ctx, int curve, int group, amdmsm_read_fn read, void *read_ctx,
                                            const void *scalars, size_t n, size_t precompute_c, size_t chunk_points,
                                            void *out_xyz, const amdmsm_opts *opts) {
    const size_t recs = amdmsm_precompute_num_digits(curve, precompute_c);
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    if (precompute_c < 2 || precompute_c > 22 || !recs) return fail(ctx, AMDMSM_ERR_BAD_ARG, "precompute_c must be 2..22");
    return stream_impl(ctx, curve, group, read, read_ctx, scalars, n, chunk_points, out_xyz, opts, recs, precompute_c);
}

int amdmsm_multi_exp_stream_with_precompute_file(amdmsm_ctx *ctx, int curve, int group, const char *path,
                                                 size_t offset_bytes, const void *scalars, size_t n,
                                                 size_t precompute_c, size_t chunk_points, void *out_xyz,
                                                 const amdmsm_opts *opts) {
    const size_t recs = amdmsm_precompute_num_digits(curve, precompute_c);
    if (!ctx) return AMDMSM_ERR_BAD_ARG;
    if (precompute_c < 2 || precompute_c > 22 || !recs) return fail(ctx, AMDMSM_ERR_BAD_ARG, "precompute_c must be 2..22");
    return stream_file_impl(ctx, curve, group, path, offset_bytes, scalars, n, chunk_points, out_xyz, opts, recs,
                            precompute_c);
}

int amdmsm_batch_exp(amdmsm_ctx *ctx, int curve, int group, size_t scalar_size, size_t window, const void *g_xyz,
                     const void *scalars, size_t n, const void *coeff, int scalars_plain, void *out_xyz) {
    GET_VT(ctx, curve, group);
    if (!g_xyz || (n && (!scalars || !out_xyz))) return fail(ctx, AMDMSM_ERR_BAD_ARG, "null pointer");
    if (window < 1 || window > 22 || scalar_size < 1 || scalar_size > (size_t)vt->fr_words * 32) {
        return fail(ctx, AMDMSM_ERR_BAD_ARG, "window / scalar_size");
    }
    const size_t xyz_bytes = (size_t)vt->el_words * 12, aff_bytes = (size_t)vt->el_words * 8, fr_bytes = (size_t)vt->fr_words * 4;
    const size_t outerc = (scalar_size + window - 1) / window;
    const size_t entries = outerc << window;
    hipStream_t st = ctx->stream;
    // Device buffers are the context's (grow-only), and the affine window table stays in HBM: a key
    // generator calls batch_exp many times with one table (libsnark r1cs_gg_ppzksnark_generator), so a
    // call with the same (group, scalar_size, window, g) as the last one skips get_window_table's work.
    fb_state &fb = ctx->fb;
    const bool same_table = fb.valid && fb.curve == curve && fb.group == group && fb.scalar_size == scalar_size &&
                            fb.window == window && memcmp(fb.g, g_xyz, xyz_bytes) == 0;
    int rc = ensure_buf(ctx, fb.small, 2 * xyz_bytes + outerc * xyz_bytes + 64);
    if (rc == AMDMSM_OK && !same_table) {
        fb.valid = false;
        rc = ensure_buf(ctx, fb.table, entries * xyz_bytes);
        if (rc == AMDMSM_OK) rc = ensure_buf(ctx, fb.table_aff, entries * aff_bytes);
    }
    if (rc == AMDMSM_OK) rc = ensure_buf(ctx, ctx->hb_sc, n ? n * fr_bytes : 16);
    if (rc == AMDMSM_OK) rc = ensure_buf(ctx, fb.out, n ? n * xyz_bytes : 16);
    if (rc) return rc;
    char *d_g = (char *)fb.small.p, *d_cf = d_g + xyz_bytes, *d_go = d_cf + xyz_bytes;
    HIP_TRY(ctx, hipEventRecord(ctx->aux_ev[0], st));
    if (!same_table) {
        HIP_TRY(ctx, hipMemcpyAsync(d_g, g_xyz, xyz_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(ctx, hipMemsetAsync(fb.table.p, 0, entries * xyz_bytes, st));   // rows shorter than 2^window: infinity beyond
    }
    if (n) HIP_TRY(ctx, hipMemcpyAsync(ctx->hb_sc.p, scalars, n * fr_bytes, hipMemcpyHostToDevice, st));
    if (coeff) HIP_TRY(ctx, hipMemcpyAsync(d_cf, coeff, fr_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(ctx->aux_ev[1], st));
    // table build and exponentiation are separate launches of the same entry: time them apart
    if (!same_table) {
        vt->fixed_base_exp(st, (const uint32_t *)d_g, (int)scalar_size, (int)window, nullptr, 0, 0, nullptr, AMDMSM_OUT_LIBFF,
                           (uint32_t *)d_go, (uint32_t *)fb.table.p, (uint32_t *)fb.table_aff.p, 1, nullptr);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipEventRecord(ctx->aux_ev[2], st));
    vt->fixed_base_exp(st, (const uint32_t *)d_g, (int)scalar_size, (int)window, (const uint32_t *)ctx->hb_sc.p, n,
                       scalars_plain ? 0 : 1, coeff ? (const uint32_t *)d_cf : nullptr, AMDMSM_OUT_LIBFF, (uint32_t *)d_go,
                       (uint32_t *)fb.table.p, (uint32_t *)fb.table_aff.p, 0, (uint32_t *)fb.out.p);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->aux_ev[3], st));
    if (n) HIP_TRY(ctx, hipMemcpyAsync(out_xyz, fb.out.p, n * xyz_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipEventRecord(ctx->aux_ev[4], st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < 4; ++i) (void)hipEventElapsedTime(&ctx->aux_ms[i], ctx->aux_ev[i], ctx->aux_ev[i + 1]);
    fb.valid = true;
    fb.curve = curve;
    fb.group = group;
    fb.scalar_size = scalar_size;
    fb.window = window;
    memcpy(fb.g, g_xyz, xyz_bytes);
    return AMDMSM_OK;
}

// device times of the last amdmsm_batch_exp of this context: ms[0] inputs host -> device, ms[1] window table
// (0 when the resident table was reused), ms[2] the exponentiations (k_fb_exp), ms[3] results device -> host
int amdmsm_get_batch_exp_timings(amdmsm_ctx *ctx, float ms[4]) {
    if (!ctx || !ms) return AMDMSM_ERR_BAD_ARG;
    std::lock_guard<std::recursive_mutex> lock(ctx->mu);
    for (int i = 0; i < 4; ++i) ms[i] = ctx->aux_ms[i];
    return AMDMSM_OK;
}

int amdmsm_batch_to_special(amdmsm_ctx *ctx, int curve, int group, void *elems_xyz, size_t stride_bytes, size_t n) {
    GET_VT(ctx, curve, group);
    const size_t xyz_bytes = (size_t)vt->el_words * 12, aff_bytes = (size_t)vt->el_words * 8;
    if (stride_bytes == 0) stride_bytes = xyz_bytes;
    if (stride_bytes != xyz_bytes) return fail(ctx, AMDMSM_ERR_BAD_ARG, "batch_to_special needs packed records");
    if (!n) return AMDMSM_OK;
    hipStream_t st = ctx->stream;
    void *d_src = nullptr, *d_aff = nullptr;
    HIP_TRY(ctx, hipMalloc(&d_src, n * xyz_bytes));
    if (hipMalloc(&d_aff, n * aff_bytes) != hipSuccess) {
        (void)hipFree(d_src);
        return fail(ctx, AMDMSM_ERR_HIP, "hipMalloc");
    }
    hipError_t e = hipMemcpyAsync(d_src, elems_xyz, n * xyz_bytes, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        vt->import_bases(st, (const uint32_t *)d_src, xyz_bytes / 4, 0, n, (uint32_t *)d_aff);
        vt->export_affine(st, (const uint32_t *)d_aff, n, (uint32_t *)d_src);
        e = hipMemcpyAsync(elems_xyz, d_src, n * xyz_bytes, hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(d_src);
    (void)hipFree(d_aff);
    if (e != hipSuccess) return fail(ctx, AMDMSM_ERR_HIP, hipGetErrorString(e));
    return AMDMSM_OK;
}

}  // extern "C"
