// Packed ("segment-interleaved") HBM layout and the multi-level partition plan.
//
// A chain of n nodes is cut into P = ceil(n/R) segments of R nodes; the LAST node of every segment is
// a separator.  One GPU lane owns one (chain b, segment p): lane = b*P + p.  For a per-node quantity
// with E doubles the element (lane, step s, e) lives at
//
//        (((lane/64)*R + s)*E + e)*64 + lane%64          (lanes padded to a multiple of 64: Lpad)
//
// i.e. [wave tile][step][element][64 lanes]: for a fixed (s, e) the 64 lanes of a wavefront read 512
// contiguous bytes, a whole step of a wavefront is one contiguous run of E*512 bytes, and a wavefront streams
// its R steps from one contiguous region -- every load and store of the sequential sweeps is fully
// coalesced while each lane still walks its own piece of the time axis in order.  The separators of level l form the chain of level l+1
// (n_{l+1} = P_l), until a level has a single segment per chain.  Levels >= 1 live in the plan workspace in the same layout, with
// their own lane count and R (mfgm_sweeps.h, coarse_off / ld_node<E, NM = true>).
#pragma once
#include <cstddef>
#include <cstdint>

namespace mfgm {

struct LevelDesc {
    int n;      // nodes per chain at this level
    int R;      // nodes per segment (last one is the separator)
    int P;      // segments per chain
    int L;      // lanes = B * P
    int Lpad;   // lanes padded to a multiple of 64
    int level;  // index of this level in its plan (0 = finest): part of the location a not-positive-definite report carries
    int nt;     // lane-per-segment plans, level 0: cache policy of the per-node arrays -- 0 default, 1 non-temporal stores, 2 loads too
                // (arrays far larger than the 256 MB of L2 + Infinity Cache are streamed: see ld_node in mfgm_sweeps.h)
};

constexpr int kMaxLevels = 8;

struct Plan {
    int B, T, d;
    int wide;      // d > 8: wavefront-per-segment kernels on natural-layout arrays (mfgm_wide.h)
    int nlevels;
    int seg_lo, seg_hi;   // level-0 segments this process owns (wide plans; the whole range unless the plan is sharded)
    // one chain over several processes (wide plans): the inputs of level `shard_level` are exchanged (summed) between the
    // processes; below it a process works on its own segments [own_lo[l], own_hi[l]) only, from it upwards everything is replicated.
    // shard_level == 0: not sharded.
    int shard_level;
    int own_lo[kMaxLevels], own_hi[kMaxLevels];
    LevelDesc lv[kMaxLevels];
    // per-level workspace offsets (in doubles) into the plan-owned level workspace, levels >= 1
    size_t off_Dhat[kMaxLevels], off_Rsub[kMaxLevels], off_S[kMaxLevels], off_rhat[kMaxLevels], off_rho[kMaxLevels];
    size_t off_L[kMaxLevels], off_G[kMaxLevels], off_y[kMaxLevels], off_Sig[kMaxLevels], off_mu[kMaxLevels];
    size_t off_part[kMaxLevels];   // per-lane partial sums (2 * Lpad doubles), all levels incl. 0
    size_t off_part2;              // second-stage scratch of the split partial sums (2 * B * 128 doubles)
    size_t off_alt;                // narrow plans with >= 2 levels: a second copy of the level-1 INPUT region [off_Dhat[1], off_L[1])
                                   // (the separator system mfgm_cq_factor_pipelined makes one factorisation ahead); 0: none
    size_t ws_doubles;             // total workspace size in doubles
};

inline size_t packed_elems(const LevelDesc& lv, int E) { return (size_t)lv.R * E * lv.Lpad; }
// kind: 0 vector, 1 full, 2 symmetric, 3 lower triangular (include/mfgm.h)
inline size_t level_elems(const Plan& P, const LevelDesc& lv, int kind) {
    const int d = P.d;
    if (P.wide) return (size_t)P.B * lv.n * (kind == 0 ? d : d * d);
    return packed_elems(lv, kind == 0 ? d : (kind == 1 ? d * d : d * (d + 1) / 2));
}

}  // namespace mfgm
