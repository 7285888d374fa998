// fused_rl.hpp -- the whole Richardson-Lucy loop of a batch of frames in ONE persistent launch
// (single view, f32, wave-private transform length; device code only).
//
// Why: in the four-launch iteration (column, ROW_RATIO, column, ROW_UPDATE: conv_kernels.hpp) the
// spectrum of every frame crosses the memory fabric at each of the four kernel boundaries: 9.7 of
// the 13.7 MB a 512^2 frame-iteration moves, against 7.34 MB algorithmic.  One frame's spectrum is
// 1.2 MB and an XCD's L2 is 4 MiB: here a TEAM of workgroups that sit on ONE XCD owns a frame for
// all K iterations, the spectrum lives in that XCD's L2, and the fabric only sees the measurement,
// the estimate and the normaliser (line_sted_tools.py:520-531 semantics unchanged; the item code
// is the four-launch path's: colconv_wave_body / row_item, so the results are bit for bit the same).
//
// Placement is OBSERVED, never assumed: every workgroup reads its XCD from HW_REG_XCC_ID, takes a
// ticket from that XCD's counter, and after a start-up barrier over the whole grid the tickets of
// an XCD are cut into teams of `team_wgs`; left-over workgroups exit.  Teams take frames
// f = team, team + n_teams, ...  No data is handed between XCDs inside the launch.
//
// Hand-off inside a team (same XCD, shared L2): plain stores (they stay in L2, dirty), every
// storing wave `s_waitcnt vmcnt(0)` (a store is counted done when L2 has it), workgroup barrier,
// one lane adds to the team's monotonic counter (agent-scope atomic) and polls it with relaxed
// agent-scope loads; readers take the spectra with `sc1` loads (L1 bypassed, L2 served) -- or,
// template parameter ACQ, through plain loads behind one `buffer_inv sc1` per workgroup and phase.
// Every spin is bounded by the 100 MHz real-time clock; a timeout raises the abort word, every
// poller sees it and the grid drains.  status[] tells the host what happened.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "conv_kernels.hpp"
#include "dev_sync.hpp"

namespace rl {

// words of the control block (unsigned; each group on its own 128-byte line).  The host zeroes
// the whole block before every launch.
enum FusedWord {
    FW_REGISTERED = 0,      // workgroups that have taken a ticket
    FW_ABORT = 32,          // != 0: a spin timed out somewhere (value = code)
    FW_FRAMES_DONE = 64,    // frames whose K iterations have completed
    FW_TEAMS = 96,          // teams formed (written by one workgroup)
    FW_XCD_COUNT = 128,     // + 32 * xcc: workgroups seen on that XCD
    FW_TEAM_BASE = 128 + 32 * 16,   // + 32 * (kFusedMaxStreams * team + stream): barrier counter of a team's stream
};
constexpr int kFusedMaxTeams = 256;
constexpr int kFusedMaxStreams = 4;
constexpr size_t kFusedCtrlWords = FW_TEAM_BASE + 32 * kFusedMaxStreams * kFusedMaxTeams;

template <typename T>
struct FusedParams {
    cx<T>* spec;            // [frames][ny][pitch]  rowFFT(est) on entry and on exit (in place)
    const T* meas;          // [frames][ny][nx]
    T* est;                 // [frames][ny][nx]
    const T* norm;          // [ny][nx]
    const cx<T>* psf_hat;   // [kx][L] transposed, pre-scaled
    const cx<T>* twy;
    const cx<T>* twx;
    unsigned* ctrl;         // kFusedCtrlWords
    int ny, nx, kx, pitch;
    int frames, iters;
    int team_wgs;           // workgroups per team
    int streams;            // frames a team keeps in flight (1 .. kFusedMaxStreams)
    unsigned timeout_us;    // per spin
    unsigned flags;         // diagnostics only (results are then wrong): 1 = never wait for a barrier, 2 = no store drain before arriving
};

// Sync policy of the fused kernel: DevSync + spectrum loads that bypass L1
// PLAIN (acquire mode): the loads are plain, L1 was invalidated at the phase boundary
template <bool PLAIN>
struct FusedSync : DevSync {
    template <typename T>
    __device__ __forceinline__ cx<T> ldg(const cx<T>* p) const {
        if constexpr (sizeof(T) == 4 && !PLAIN) {
            const unsigned long long u =
                __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return mk<T>(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
        } else {
            return *p;
        }
    }
};

__device__ __forceinline__ unsigned fused_ld(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned fused_add(unsigned* p, unsigned v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void fused_st(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one lane: wait until *word >= target.  false: timed out or the grid is aborting.
__device__ __forceinline__ bool fused_wait_ge(unsigned* ctrl, const unsigned* word, unsigned target, unsigned timeout_us, unsigned code) {
    if (fused_ld(word) >= target) return true;   // the common case when the wait was hidden behind other work
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    for (unsigned n = 1;; ++n) {
        __builtin_amdgcn_s_sleep(1);
        if (fused_ld(word) >= target) return true;
        if ((n & 15) == 0) {   // the abort word and the clock are looked at every 16th poll
            if (fused_ld(ctrl + FW_ABORT) != 0) return false;
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)timeout_us * 100ull) {
                fused_st(ctrl + FW_ABORT, code);
                return false;
            }
        }
    }
}

// S: frames a team keeps in flight.  With S > 1 the team walks the phases of its S frames in turn
// (frame A phase p, frame B phase p, A phase p+1, ...): a workgroup ARRIVES at a frame's barrier when its
// items of that phase are stored, works on the other frames' phases, and only then WAITS for that
// barrier -- the barrier latency and the arrival skew of a phase hide behind the other frames' items.
template <class Cfg, int NW, bool ACQ, typename T>
__device__ __forceinline__ void fused_rl_body(const FusedParams<T>& p, cx<T>* lds, int* sh) {
    static_assert(Cfg::T == 64, "fused kernel needs wave-private transforms");
    constexpr int LP = LdsSlots<Cfg>::value;
    const int tid = (int)threadIdx.x;
    const int wave = rl_uniform(tid / 64);
    const unsigned lane = (unsigned)(tid % 64);
    const int W = p.team_wgs;
    const int S = p.streams < 1 ? 1 : (p.streams > kFusedMaxStreams ? kFusedMaxStreams : p.streams);

    // ---- team formation ----
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 15u;
        const unsigned ticket = fused_add(p.ctrl + FW_XCD_COUNT + 32 * xcc, 1u);
        fused_add(p.ctrl + FW_REGISTERED, 1u);
        int team = -1, member = 0, n_teams = 0;
        if (fused_wait_ge(p.ctrl, p.ctrl + FW_REGISTERED, gridDim.x, p.timeout_us, 1u)) {
            unsigned before = 0, mine = 0;
            for (unsigned x = 0; x < 16; ++x) {
                const unsigned t = fused_ld(p.ctrl + FW_XCD_COUNT + 32 * x) / (unsigned)W;
                if (x < xcc) before += t;
                if (x == xcc) mine = t;
                n_teams += (int)t;
            }
            if (n_teams > kFusedMaxTeams) n_teams = -1;   // cannot happen with a sane grid; refuse
            if (n_teams > 0 && ticket < mine * (unsigned)W) {
                team = (int)(before + ticket / (unsigned)W);
                member = (int)(ticket % (unsigned)W);
                if (team == 0 && member == 0) fused_st(p.ctrl + FW_TEAMS, (unsigned)n_teams);
            }
        }
        sh[0] = team;
        sh[1] = member;
        sh[2] = n_teams;
    }
    __syncthreads();
    const int team = sh[0], member = sh[1], n_teams = sh[2];
    if (team < 0) return;
    unsigned* const counters = p.ctrl + FW_TEAM_BASE + 32 * kFusedMaxStreams * team;   // + 32 * stream

    FusedSync<ACQ> sync;

    ColParams<T> cp;
    cp.in = p.spec; cp.out = p.spec; cp.psf_hat = p.psf_hat; cp.tw = p.twy;
    cp.ny = p.ny; cp.kx = p.kx; cp.pitch = p.pitch; cp.V = 1; cp.in_sb = 1; cp.in_sv = 0;
    cp.mode = COL_PER_IMAGE; cp.images = p.frames; cp.order = 1;
    RowParams<T> rp;
    rp.spec_in = p.spec; rp.spec_out = p.spec; rp.src = p.meas; rp.dst = p.est; rp.norm = p.norm; rp.scale = nullptr;
    rp.tw = p.twx; rp.ny = p.ny; rp.nx = p.nx; rp.pitch = p.pitch; rp.V = 1; rp.frames = p.frames;

    const int tiles = (p.kx + NW - 1) / NW;
    const int pairs = (p.ny + 1) / 2;

    // this workgroup's items of a phase are stored: count it in on the stream's counter
    auto arrive = [&](int s) {
        if (!(p.flags & 2u)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores are in L2
        __syncthreads();
        if (tid == 0) fused_add(counters + 32 * s, 1u);
    };
    // every member has made `arrivals` arrivals on the stream's counter (every workgroup arrives once per
    // phase of the stream, with or without items); false: the grid is aborting.  A wait must be followed
    // by an arrive (a workgroup barrier) before the next wait, or by resync.
    auto wait = [&](int s, unsigned arrivals, unsigned code, bool resync) -> bool {
        if (tid == 0) {
            const bool ok = fused_wait_ge(p.ctrl, counters + 32 * s, arrivals * (unsigned)W, p.timeout_us, code);
            if (ok && ACQ) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: this CU's L1
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            sh[3] = ok ? 1 : 0;
        }
        __syncthreads();
        const bool ok = sh[3] != 0;
        if (resync) __syncthreads();   // sh[3] is read before the next wait rewrites it
        return ok;
    };

    // Everything loop invariant -- twiddle loads, per-lane addresses -- is kept INSIDE the item code
    // (stream_launder*: opaque to the optimiser), or the compiler hoists it out of the iteration loop
    // into ~140 spilled registers.
    // rot: rotates which members take the left-over tiles of a phase (37 tiles on 32 workgroups)
    auto col_phase = [&](int frame, int rot) {
        for (int tile = (member + W - rot % W) % W; tile < tiles; tile += W) {
            ColParams<T> c = cp;
            c.tw = stream_launder_ptr(cp.tw);
            c.psf_hat = stream_launder_ptr(cp.psf_hat);
            colconv_wave_body<Cfg, NW, COL_PER_IMAGE, T>(c, stream_launder_lane(tid), tile, frame, lds, sync);
            __syncthreads();   // the tile is stored before the next one is loaded into the same LDS
        }
    };
    // row pairs are dealt to the waves of the team (wave-major over the members), so that a team
    // larger than the row phase needs spreads the active waves over all its workgroups
    auto row_phase = [&](int frame, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        for (int pr = wave * W + member; pr < pairs; pr += W * NW) {
            RowParams<T> r = rp;
            r.tw = stream_launder_ptr(rp.tw);
            const int ln = stream_launder_lane((int)lane);
            RowSpectra<Cfg, T> in;
            in.request(r, frame, 2 * pr, (unsigned)ln, sync);
            row_item<Cfg, MODE, false>(r, (unsigned)ln, ln, frame, 2 * pr, in, LdsView<T, 1, LdsGather<Cfg::L>::value>{lds + wave * LP},
                                       r.tw, sync, [] {});
        }
    };
    using RatioTag = std::integral_constant<int, ROW_RATIO>;
    using UpdateTag = std::integral_constant<int, ROW_UPDATE>;

    int rot = 0;
    unsigned before = 0;   // arrivals on each (active) stream's counter in earlier frame groups
    for (int base = team * S; base < p.frames; base += n_teams * S, before += 4u * (unsigned)p.iters) {
        // line_sted_tools.py:520-531 per frame: 0 H(est) columns, 1 ratio rows, 2 H_t columns, 3 update rows
#pragma unroll 1
        for (int step = 0; step < 4 * p.iters; ++step) {
            const int ph = step & 3;
#pragma unroll 1
            for (int s = 0; s < S; ++s) {
                const int frame = base + s;
                if (frame >= p.frames) break;
                // the frame's previous phase is complete (a stream's frames of earlier groups are done:
                // every stream that is active now was active in all earlier groups)
                if (step > 0 && !(p.flags & 1u) && !wait(s, before + (unsigned)step, 2u + (unsigned)ph, false)) return;
                if (ph == 0 || ph == 2) col_phase(frame, rot);
                else if (ph == 1) row_phase(frame, RatioTag{});
                else row_phase(frame, UpdateTag{});
                arrive(s);
                rot += 5;
            }
        }
        if (member == 0 && p.iters > 0) {   // the frames are done when every member has arrived at their last barriers
            unsigned done = 0;
            for (int s = 0; s < S && base + s < p.frames; ++s) {
                if (!wait(s, before + 4u * (unsigned)p.iters, 6u, true)) return;
                ++done;
            }
            if (tid == 0) fused_add(p.ctrl + FW_FRAMES_DONE, done);
        }
    }
}

}  // namespace rl
