// ocs_device_common.hpp -- device helpers shared by the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>

namespace ocs {

// The wave that carries a kernel's dependent chain (state / costate recursion) may ask the SIMD's arbiter for priority
// over the helper waves it shares the SIMD with (OCS_CHAIN_PRIO: s_setprio level).  Off: measured with level 3 on the
// headline pass pair and the folded sweep (scripts/ab_lib.sh) -- state pass 58.9-60.1 -> 60.0-60.7 us, sweep 177.7-178.6
// -> 179.1-179.4 us: the chain waits for its own results (8 dependent fp64 operations per step), not for issue slots.
#ifndef OCS_CHAIN_PRIO
#define OCS_CHAIN_PRIO 0
#endif
__device__ static inline void chain_wave_priority() {
  if (OCS_CHAIN_PRIO) __builtin_amdgcn_s_setprio(OCS_CHAIN_PRIO);
}

#define OCS_INLINE __attribute__((always_inline))

// First trajectory of a workgroup's tile of TPW trajectories.  A batch that is not a multiple of the tile: the LAST workgroup takes
// the last TPW trajectories, overlapping its neighbour -- the overlap is computed twice with the same operations and stored twice
// with the same values (the tiled kernels accumulate nothing across trajectories; counters of active instances may count an
// instance twice, they are only compared with zero).  Launchers ask for it with batch > TPW and an even batch (16-byte DMA chunks).
__device__ static inline int tile_base(int block, int TPW, int batch) {
  const int b0 = block * TPW;
#ifdef OCS_NO_TILE_OVERLAP   // (A/B builds)
  return b0;
#endif
  return b0 + TPW <= batch ? b0 : batch - TPW;
}

// Wave-uniform read-only tables (step sizes, time coefficients, shared parameters) are read
// through the constant address space so the backend emits scalar (s_load) instead of
// per-lane vector loads: `__restrict__` on struct members does not reach alias analysis.
typedef const double __attribute__((address_space(4))) * uniform_ptr;
__device__ static inline uniform_ptr as_uniform(const double* p) { return (uniform_ptr)p; }
#define PSU(p) as_uniform(p)

// Where a trajectory's parameters come from: the shared block `ps` (uniform, scalar loads) or, for the
// parameters whose bit is set in `pmask`, the per-trajectory array pb[k][batch].
struct ParamSrc {
  uniform_ptr ps;
  const double* pb;
  unsigned pmask;
  size_t B;
  int b;
  __device__ inline double operator()(int k) const {
    return ((pmask >> k) & 1u) ? pb[(size_t)k * B + b] : ps[k];
  }
};

// ---------------------------------------------------------------------------------------
// per-step record table
// ---------------------------------------------------------------------------------------
// Everything wave-uniform that step i needs sits in one 64-byte-aligned record:
//   REC[i] = { h, h/2, h/6, h/3, tc(t_2i)[NTC], tc(t_2i+1)[NTC], tc(t_2i+2)[NTC], pad | sc[8] }
__host__ __device__ constexpr int rec_sc_offset(int ntc) { return ((4 + 3 * ntc + 7) / 8) * 8; }
// ... followed by a block of 8 problem-defined step constants (P::step_consts; used by the pipeline kernels)
__host__ __device__ constexpr int rec_stride(int ntc) { return rec_sc_offset(ntc) + 8; }
// The table carries kRecPad extra records before step 0 and after step N-1 (copies of the edge
// records) so that the kernels can keep a ring of prefetched records in flight by just walking a
// pointer, without clamping the index at either end.
constexpr int kRecPad = 4;

template <int NTC>
struct StepRec {
  double h, hh, h6, h3;
  double tcA[NTC], tcM[NTC], tcB[NTC];
};
// Records are read through the VECTOR memory path (every lane loads the same address): vector
// loads return in order, so a ring of RD records can be kept in flight with counted vmcnt
// waits.  Scalar loads return out of order, every wait is lgkmcnt(0), and the newest
// prefetch would always be waited for together with the record that is needed (measured:
// +40 % time per step).
template <int NTC>
__device__ static inline StepRec<NTC> load_rec(const double* q) {
  StepRec<NTC> r;
  r.h = q[0];
  r.hh = q[1];
  r.h6 = q[2];
  r.h3 = q[3];
#pragma unroll
  for (int k = 0; k < NTC; ++k) {
    r.tcA[k] = q[4 + k];
    r.tcM[k] = q[4 + NTC + k];
    r.tcB[k] = q[4 + 2 * NTC + k];
  }
  return r;
}

// pchip (Fritsch-Carlson / MATLAB pchipslopes) interior slope of a node between secants del0, del1 with weights w1, w2:
// with ONE division: |del0 del1| / (w1 |del0| + w2 |del1|) (pchip_interior's harmonic mean multiplied
// through by dmax; round-off level difference to the two-division form of the midpoint kernel)
__device__ static inline double pchip_interior1(double del0, double del1, double w1, double w2) {
  const bool same = (del0 > 0.0 && del1 > 0.0) || (del0 < 0.0 && del1 < 0.0);
  const double a0 = fabs(del0), a1 = fabs(del1);
  double d = (a0 * a1) / __builtin_fma(w1, a0, w2 * a1);
  asm("" : "+v"(d));  // opaque: keeps the division out of a divergent branch (0/0 lanes are discarded below); not
                      // volatile, so that the divisions of neighbouring nodes still overlap
  return same ? (del0 > 0.0 ? d : -d) : 0.0;
}

// the same slope from the signed secants: del0 del1 / (w1 del0 + w2 del1) -- the quotient of the absolute values with the
// common sign of the secants, bit for bit, in fewer instructions (k_forward_cc's control waves are VALU-bound)
__device__ static inline double pchip_interior_s(double del0, double del1, double w1, double w2) {
  const double pr = del0 * del1;
  double d = pr / __builtin_fma(w1, del0, w2 * del1);
  asm("" : "+v"(d));
  return pr > 0.0 ? d : 0.0;
}

// ... and with the quotient by reciprocal + Newton steps instead of the IEEE division sequence (no scaling: the secants of
// a costate are far from the ends of the exponent range; at most an ulp from the correctly rounded quotient)
// (v_rcp_f64 is good to 2^29 ulp, i.e. ~1e-7 relative; one Newton step on the reciprocal takes that to ~1e-14, and the
//  correction of the quotient multiplies the two errors: full precision in six operations)
__device__ static inline double fast_div(double n, double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  const double q = n * r;
  return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
__device__ static inline double pchip_interior_f(double del0, double del1, double w1, double w2) {
  const double pr = del0 * del1;
  const double d = fast_div(pr, __builtin_fma(w1, del0, w2 * del1));
  return pr > 0.0 ? d : 0.0;
}

// per-interval pchip records of the node grid (built by k_pchip_records, streamed by k_costate_plx / k_forward_cc):
// doubles per interval i: {h(i-1), h(i), h(i+1), 1/h(i-1), 1/h(i), 1/h(i+1), W1(i), W2(i), W1(i+1), W2(i+1), tmid_i - t_i, h(i)/8, pad}
// pchip at the MIDDLE of an interval (all the sweep kernels need) is the Hermite cubic's closed form there,
// (y0 + y1)/2 + h/8 (d0 - d1): four operations instead of the ten of the general evaluation; tmid_i is the rounded
// (t_i + t_i+1)/2, so this differs from evaluating at tmid_i by round-off
constexpr int kPRec = 16;
// pchip end slope (MATLAB pchipslopes' three-point formula with its two shape-preserving corrections)
__device__ static inline double pchip_end_pl(double h0, double h1, double del0, double del1) {
  double d = ((2.0 * h0 + h1) * del0 - h0 * del1) / (h0 + h1);
  const bool s0 = (d > 0.0) == (del0 > 0.0) && (d < 0.0) == (del0 < 0.0);
  const bool s1 = (del0 > 0.0) == (del1 > 0.0) && (del0 < 0.0) == (del1 < 0.0);
  if (!s0)
    d = 0.0;
  else if (!s1 && fabs(d) > fabs(3.0 * del0))
    d = 3.0 * del0;
  return d;
}

// The record table is written by another kernel and is cold in this XCD's L2: a scalar load
// that misses to the Infinity Cache / HBM costs more than a whole RK4 step.  Every wave
// therefore sweeps the table once with wide vector loads (1 KiB per instruction) before the
// recursion starts, so the per-step scalar loads that follow are L2 hits.
__device__ static inline double warm_table(const double* tab, size_t ndoubles) {
  typedef double double2v __attribute__((ext_vector_type(2)));
  const double2v* q = reinterpret_cast<const double2v*>(tab);
  const size_t n2 = ndoubles / 2;
  double acc = 0.0;
  size_t k = threadIdx.x & 63;
  for (; k + 15 * 64 < n2; k += 16 * 64) {
    double2v v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = q[k + (size_t)j * 64];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc += v[j].x + v[j].y;
  }
  for (; k < n2; k += 64) acc += q[k].x + q[k].y;
  return acc;  // the caller keeps it alive behind a never-taken branch
}


}  // namespace ocs
