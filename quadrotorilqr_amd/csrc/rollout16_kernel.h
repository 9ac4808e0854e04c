// rollout16_kernel.h -- k_rollout16: sixteen lanes per trajectory, four trajectories per block (up to 4096 trajectories); its body is
// rollout16_body.inc (k_backward_rollout and k_round contain it too); the lane-level arithmetic is rollout16.h.
// Part of the device code of libquadrotor_ilqr.so (gfx950 only); ilqr_kernels.h includes every part.
#pragma once

#include "kernels_common.h"

namespace qilqr {

// ---------------------------------------------------------------------------------------------
// k_rollout16: SIXTEEN LANES PER TRAJECTORY (rollout16.h): block = 192 = control wavefront A + pose wavefront B +
// operand wavefront P for four trajectories (b0 = 4 blockIdx.x; row r of 16 lanes <-> trajectory b0 + r).  At B = 1024
// that is 256 blocks -- one per CU -- instead of the 16 blocks of k_rollout3, and about a third of its instructions
// per knot.  The recurrence spans two knots (Log_i -> u_i -> v_{i+1} -> E_{i+1} -> T_{i+2} -> Log_{i+2}), so two knots can be
// in flight, and they are given to two IDENTICAL wavefronts one knot apart rather than to roles:
//   X_p (r16_wave_X; p = 0, 1): the whole step of the knots of parity p -- tau_i = Log(T_nom^-1 T_i),
//      u_i = u_nom + alpha k + K [tau_i ; v_i - v_nom], v_{i+1} = v_i + dt a(q_i, v_i, u_i), E_{i+1} = Exp(dt v_{i+1}),
//      T_{i+2} = T_{i+1} E_{i+1} -- from its own T_i and the other wave's v_i and T_{i+1}; stores knot i
//   P (r16::p_load, p_compute): for knot k (running up to R16_RING - 1 knots ahead): loads the nominal knot and the
//      gains (tiled global layout, per-lane element offsets), forms the 23 operand registers and writes them to ring slot
//      k % R16_RING as [register][lane]; copies the time column to the output trajectory.
// The waves never meet at a barrier inside the loop.  LDS words carry progress: knots produced by P, "v_k ready" (which
// also frees knot k - 1's operand slot), "T_k ready"; the values themselves go through four-deep LDS slots (k & 3).
// LDS operations of a wavefront execute in order, so a flag written after the data is seen after the data; every spin
// is bounded, so a lost flag ends the kernel instead of hanging it.
// Trajectories of the block that are not being rolled out this round alias the block's first live trajectory (their
// rows compute a duplicate that nobody stores): no row wanders onto a slow branch, no extra memory traffic.
// S = storage precision of trajectories and gains; the arithmetic is fp64 in either mode.
// ---------------------------------------------------------------------------------------------
// the sixteen-term series of Exp's four coefficients (se3_math.h, Series::exp2) in LDS: [place in the row][coefficient]; rollout16_body.inc
// fills it in front of the kernel's first barrier
__shared__ double r16_exp2_lds[4][16];
struct DevWave {
  typedef double V;
  typedef bool M;
  typedef int I;
  template <class F> static __device__ __forceinline__ V vconst(F f) { return f((int)(threadIdx.x & 63)); }
  template <class F> static __device__ __forceinline__ M mconst(F f) { return f((int)(threadIdx.x & 63)); }
  // element indices of a knot become offsets into the tiled layout, (e / 2) * TILE2 + e % 2 (se3_math.h, knot_elem): the
  // loads of wavefront P are then a wave-uniform knot pointer plus a 32-bit lane offset, no address arithmetic per knot
  static __device__ __forceinline__ I iuni(int e) { return (e >> 1) * TILE2 + (e & 1); }
  template <class F> static __device__ __forceinline__ I iconst(F f) { return iuni(f((int)(threadIdx.x & 63))); }
  // value of lane L of the caller's row of 16 (v_mov_b64_dpp row_newbcast)
  template <int L> static __device__ __forceinline__ V bc(V x) { return __builtin_amdgcn_mov_dpp(x, 0x150 + L, 0xf, 0xf, false); }
  // acc + x[lane L of the row] * m
  template <int L> static __device__ __forceinline__ V fm(V acc, V src, V m) {
    return __builtin_fma(bc<L>(src), m, acc);
  }
  // acc + sum_c src[lane L0 + c of the row] * m_c as a chain of v_fmac_f64_dpp (one instruction per term; the compiler
  // itself emits v_mov_b64_dpp + v_fma_f64, two).  The compiler's hazard recogniser does not look inside the asm, so the block
  // carries its own wait states on both sides: a DPP read needs two behind the VALU write of its source -- `src` may have
  // just been written, and `acc` may be the source of a DPP read right after.
#define QILQR_FMAC_DPP(m, l) "v_fmac_f64_dpp %0, %1, " m " row_newbcast:" l " row_mask:0xf bank_mask:0xf\n\t"
  template <int L0> static __device__ __forceinline__ V dot2(V acc, V src, V m0, V m1) {
    asm("s_nop 1\n\t" QILQR_FMAC_DPP("%2", "%4") QILQR_FMAC_DPP("%3", "%5") "s_nop 1"
        : "+v"(acc) : "v"(src), "v"(m0), "v"(m1), "n"(L0), "n"(L0 + 1));
    return acc;
  }
  template <int L0> static __device__ __forceinline__ V dot3(V acc, V src, V m0, V m1, V m2) {
    asm("s_nop 1\n\t" QILQR_FMAC_DPP("%2", "%5") QILQR_FMAC_DPP("%3", "%6") QILQR_FMAC_DPP("%4", "%7") "s_nop 1"
        : "+v"(acc) : "v"(src), "v"(m0), "v"(m1), "v"(m2), "n"(L0), "n"(L0 + 1), "n"(L0 + 2));
    return acc;
  }
  template <int L0> static __device__ __forceinline__ V dot4(V acc, V src, V m0, V m1, V m2, V m3) {
    asm("s_nop 1\n\t" QILQR_FMAC_DPP("%2", "%6") QILQR_FMAC_DPP("%3", "%7") QILQR_FMAC_DPP("%4", "%8") QILQR_FMAC_DPP("%5", "%9") "s_nop 1"
        : "+v"(acc) : "v"(src), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "n"(L0), "n"(L0 + 1), "n"(L0 + 2), "n"(L0 + 3));
    return acc;
  }
#undef QILQR_FMAC_DPP
  // permutation inside every quad of four lanes (two v_mov_b32_dpp quad_perm: fp64 DPP has row_newbcast only)
  template <int CTRL> static __device__ __forceinline__ V qperm(V x) {
    const long long v = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_mov_dpp((int)(v >> 32), CTRL, 0xf, 0xf, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
  }
  static __device__ __forceinline__ V rot1(V x) { return qperm<0xC9>(x); }  // lane j <- j + 1 (mod 3), j = 3 stays
  static __device__ __forceinline__ V fma(V a, V b, V c) { return __builtin_fma(a, b, c); }
  static __device__ __forceinline__ bool any(M m) { return __ballot(m) != 0ull; }
  static __device__ __forceinline__ V sel(M m, V a, V b) { return m ? a : b; }
  static __device__ __forceinline__ M gt(V a, V b) { return a > b; }
  static __device__ __forceinline__ M lt(V a, V b) { return a < b; }
  static __device__ __forceinline__ M land(M a, M b) { return a && b; }
  static __device__ __forceinline__ M lor(M a, M b) { return a || b; }
  static __device__ __forceinline__ M lnot(M a) { return !a; }
  static __device__ __forceinline__ V abs_(V a) { return fabs(a); }
  static __device__ __forceinline__ V exp2_coeff(int k) { return r16_exp2_lds[threadIdx.x & 3][k]; }
  static __device__ __forceinline__ V sqrt_(V a) { return sqrt(a); }
  static __device__ __forceinline__ V sin_(V a) { return sin(a); }
  static __device__ __forceinline__ V cos_(V a) { return cos(a); }
  static __device__ __forceinline__ V atan2_(V a, V b) { return atan2(a, b); }
  // the closed forms beyond the series' ranges: rarely taken and large (sin, cos, atan2 in fp64), so out of line -- the knot
  // loops stay short and the closed forms' registers are not the loops'
  static __device__ __attribute__((noinline)) V exp_closed(M c, V x, V p, M l0, M l1, M l2, M l3) {
    return r16::exp_closed_forms<DevWave>(c, x, p, l0, l1, l2, l3);
  }
  static __device__ __attribute__((noinline)) V log_closed(M c, V s2, V wq, V coeff) { return r16::log_closed_forms<DevWave>(c, s2, wq, coeff); }
  static __device__ __attribute__((noinline)) V jinv_closed(M c, V th2, V cJ) { return r16::jinv_closed_forms<DevWave>(c, th2, cJ); }
};

// s_waitcnt vmcnt(0) as an instruction the compiler's wait-count pass sees (gfx9 encoding: vmcnt in bits 3:0 and 15:14,
// expcnt 6:4 = 7 and lgkmcnt 11:8 = 15 left open).  Placed after the loads of a role's initial state: otherwise the pass may
// keep "a load is outstanding" alive around the knot loop and wait for vmcnt(0) INSIDE it -- which, the counter being shared,
// also waits for the knot's own stores, every knot.
#define R16_LOADS_DONE() __builtin_amdgcn_s_waitcnt(0x0F70)
constexpr int R16_RING = 4;
constexpr int R16_SPIN_MAX = 1 << 22;
#ifdef QILQR_DIAG
// diagnostics build: the knot whose velocity hand-off a step wavefront withholds (-1: none), so that the other wavefront's
// bounded spin runs out and the block's abort path is taken (tests/test_gpu_robustness.py)
__device__ int g_r16_stall_knot = -1;
#endif
constexpr int R16_CHUNK = 16;  // knots per "stored and visible" announcement of wave A (k_solve4's linearisation follows it)
enum { R16_F_PROD = 0, R16_F_V, R16_F_T, R16_F_K0, R16_F_K1, R16_F_ABORT, R16_F_TASK, R16_NFLAGS };  // (R16_F_TASK: k_round's followers, round_kernels.h)
enum { X_V = 0, X_T = 1 };
// LDS of the three rollout roles
struct R16Lds {
  double ops[R16_RING][r16::NOPS][64];  // operand registers of R16_RING knots, [register][lane]
  double xch[2][4][2][64];             // hand-off slots [X_V (v_lin, omega) of knot k | X_T (t, q) of knot k][k & 3][register][lane]
  int flags[R16_NFLAGS];               // knots produced by P; v_k ready (and knot k - 1's operands used); T_k ready; even, odd knots stored; abort
};
// The LDS executes the operations of one wavefront in the order they were issued, so a flag written after the data (or
// after the reads of a slot) is seen after them: no s_waitcnt, and no workgroup fence -- a release fence would wait for the
// wavefront's outstanding GLOBAL loads and stores too (vmcnt(0)), i.e. for P's prefetch and the knot stores, on every
// knot.  The asm statements only keep the compiler from moving LDS accesses across the flag.
__device__ __forceinline__ int r16_flag_read(R16Lds &sh, int which) {
  return __hip_atomic_load(&sh.flags[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void r16_flag_post(R16Lds &sh, int which, int value, int lane) {
  asm volatile("" ::: "memory");
  if (lane == 0) __hip_atomic_store(&sh.flags[which], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// false: the flag never came (a bounded spin: the kernel ends instead of hanging); the abort word tells the other roles
__device__ __forceinline__ bool r16_flag_wait(R16Lds &sh, int which, int target, int seen, int lane) {
  if (__builtin_expect(seen >= target, 1)) return true;  // already observed (read ahead, one knot ago)
  for (int spins = 0; spins < R16_SPIN_MAX; ++spins) {
    if (r16_flag_read(sh, which) >= target) {
      asm volatile("" ::: "memory");
      return true;
    }
    if ((spins & 255) == 255 && r16_flag_read(sh, R16_F_ABORT)) break;
  }
  r16_flag_post(sh, R16_F_ABORT, 1, lane);
  return false;
}
// the same for a wavefront that is in no hurry (the linearisation waits a chunk of knots at a time): sleeps between polls
__device__ __forceinline__ bool r16_flag_wait_relaxed(R16Lds &sh, int which, int target, int lane) {
  for (int spins = 0; spins < R16_SPIN_MAX; ++spins) {
    if (r16_flag_read(sh, which) >= target) {
      asm volatile("" ::: "memory");
      return true;
    }
    if (r16_flag_read(sh, R16_F_ABORT)) return false;
    __builtin_amdgcn_s_sleep(4);
  }
  r16_flag_post(sh, R16_F_ABORT, 1, lane);
  return false;
}
// A hand-off read in ONE LDS round trip: the flag and the NV values are requested back to back (the LDS serves a
// wavefront's requests in order, so values read after a flag that shows `target` are the published ones); if the flag is not
// there yet, poll and read again.
template <int NV>
__device__ __forceinline__ bool r16_read_handoff(R16Lds &sh, int which, int target, int kind, int par, double (&d)[NV], int lane) {
  for (int spins = 0; spins < R16_SPIN_MAX; ++spins) {
    asm volatile("" ::: "memory");  // read again, every time round
    const int f = r16_flag_read(sh, which);
    double a[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) a[r] = sh.xch[kind][par][r][lane];
    if (__builtin_expect(f >= target, 1)) {
#pragma unroll
      for (int r = 0; r < NV; ++r) d[r] = a[r];
      asm volatile("" ::: "memory");
      return true;
    }
    if ((spins & 255) == 255 && r16_flag_read(sh, R16_F_ABORT)) break;
  }
  r16_flag_post(sh, R16_F_ABORT, 1, lane);
  return false;
}
// the first attempt of r16_read_handoff split off, so that its LDS latency can be covered by other work: request the flag
// and the values here, do the other work, then r16_handoff_finish (which polls only if the first attempt came too early)
template <int NV>
__device__ __forceinline__ void r16_handoff_request(R16Lds &sh, int which, int kind, int par, int &f, double (&a)[NV], int lane) {
  asm volatile("" ::: "memory");
  f = r16_flag_read(sh, which);
#pragma unroll
  for (int r = 0; r < NV; ++r) a[r] = sh.xch[kind][par][r][lane];
  asm volatile("" ::: "memory");
}
template <int NV>
__device__ __forceinline__ bool r16_handoff_finish(R16Lds &sh, int which, int target, int kind, int par, int f, double (&a)[NV], int lane) {
  if (__builtin_expect(f >= target, 1)) return true;
  return r16_read_handoff<NV>(sh, which, target, kind, par, a, lane);
}
// "my knots up to i are stored and visible to the block": every fourth knot of its own a step wave waits for ALL its
// outstanding vector-memory operations (s_waitcnt vmcnt(0)) and announces the knots it has stored so far.  (Round 2 waited
// for all but the eight youngest operations -- vmcnt(8), "the stores of knots i .. i - 6 may be in flight" -- which is only
// right while a step issues exactly two stores and nothing else that counts: a spill inside the loop would have made the
// announcement early and the followers read knots not yet written, silently.  The full wait costs one store latency per
// eight knots on a path that only k_solve4 takes.)
// PUBLISH = 2 (k_round, whose other wavefronts linearise the candidate's knots behind the rollout, a chunk of sixteen at a time:
// round_kernels.h): the wavefront's first knot of a chunk announces its last knot of the chunk before -- all vector-memory operations but the
// two youngest are complete (s_waitcnt vmcnt(2): a step's two stores are
// the only ones the loop issues; memory operations of one kind complete in order; anything else the compiler might add there -- a spill --
// would be younger still and only make the wait longer, never the announcement early), which stalls for nothing in the steady state.
template <int PUBLISH>
__device__ __forceinline__ void r16_publish_stores(R16Lds &sh, int which, int i, int last, int lane) {
  if (PUBLISH == 2) {
    if (i == last) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      r16_flag_post(sh, which, i + 1, lane);
    } else if (i >= R16_CHUNK && (i & (R16_CHUNK - 2)) == 0) {  // (i = 16 k or 16 k + 1: the wavefront's knot i - 2 completes a chunk of sixteen)
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      r16_flag_post(sh, which, i - 1, lane);
    }
    return;
  }
  if (i == last) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    r16_flag_post(sh, which, i + 1, lane);
  } else if (((i >> 1) & 3) == 3 && i >= 16) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    r16_flag_post(sh, which, i + 1, lane);
  }
}

// P: operand registers.  traj / gains: the nominal trajectory and the gains of the lane's trajectory (tiled); out: its
// candidate trajectory (the time column is copied there when `live`).
template <typename S>
__device__ __forceinline__ void r16_wave_P(R16Lds &sh, const S *traj, const S *gains, S *out, double alpha, bool live, int n, int lane,
                                           unsigned long long *stamps_out) {
  using namespace r16;
  PConsts<DevWave> pc;
  make_pconsts(pc);
  // The loads of knot k + 1 are requested before knot k is converted and written (two register sets, loop unrolled
  // by two: no copies).  A knot is 30 loads per lane: a wave-uniform knot pointer plus the lane's 32-bit offset.
  // The requests are unconditional (past the end the last knot is requested again): a branch around them makes the
  // compiler wait for vmcnt(0) at every use, i.e. for the requests it has just issued.
  S rawA[NRAW], rawB[NRAW], tmA, tmB;
  auto request = [&](int k, S (&raw)[NRAW], S &tm) {
    const int kk = k < n ? k : n - 1;
    const S *tk = traj + (long)kk * (9 * TILE2), *gk = gains + (long)kk * (26 * TILE2);
    tm = tk[0];  // time_s: the oldest request of the knot
    auto ld = [&](int off) -> S { return tk[off]; };
    auto lg = [&](int off) -> S { return gk[off]; };
    p_load<DevWave>(pc, ld, lg, raw);
  };
  bool ok = true;
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  auto knot = [&](int k, S (&rc)[NRAW], S &tmc, S (&rn)[NRAW], S &tmn) {
    request(k + 1, rn, tmn);
    QSTAMP(0);  // P: requests
    double op[NOPS];
    p_compute<DevWave>(pc, rc, alpha, op);
    QKEEP(op[0]); QKEEP(op[r16::NOPS - 1]); QKEEP(op[10]);
    QSTAMP(1);  // P: wait for the loads, operand registers
    // slot k % R16_RING is free once the step of knot k - R16_RING has used its operands: that step posts v_{k - R16_RING + 1}
    // behind their last use, and the "v ready" word only grows (a step posts after it has taken the previous step's v)
    if (k >= R16_RING && !r16_flag_wait(sh, R16_F_V, k - R16_RING + 1, -1, lane)) ok = false;
    QSTAMP(2);  // P: wait for a free slot
#pragma unroll
    for (int r = 0; r < NOPS; ++r) sh.ops[k % R16_RING][r][lane] = op[r];
    r16_flag_post(sh, R16_F_PROD, k + 1, lane);
    if (live && (lane & 15) == 0) out[knot_elem<true>(k, 0, 18)] = tmc;  // time_s passes through (ilqr.hh:164)
    QSTAMP(3);  // P: LDS writes, flag, time store
  };
  request(0, rawA, tmA);
  for (int k = 0; k < n && ok; k += 2) {
    knot(k, rawA, tmA, rawB, tmB);
    if (k + 1 < n && ok) knot(k + 1, rawB, tmB, rawA, tmA);
  }
#ifdef QILQR_STAMPS
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
}

// X_p: the steps of the knots of parity p.  Step i, from T_i (this wave's own, out of its step i - 2):
//     tau_i = Log(T_nom^-1 T_i);  u_i;  v_{i+1} = v_i + dt a(q_i, v_i, u_i)  -> handed to the other wave;
//     E_{i+1} = Exp(dt v_{i+1});  T_{i+2} = T_{i+1} E_{i+1}                   -> handed to the other wave, and kept
// with v_i and T_{i+1} from the other wave's step i - 1.  The recurrence spans two knots, so two such waves, one knot apart,
// never wait for each other in the steady state: v_i is posted about half a step before step i needs it, T_{i+1} likewise --
// the hand-offs' LDS latency (~450 cycles from post to use through a progress word, which bounded every partition of a
// knot into roles: 74 us) is off the chain.  (TT, QQ, VL, VW): the state of knot 0.  PUBLISH: announce the stored knots (1: a chunk at a time, k_solve4; 2: every knot, k_round).
template <typename S, int PUBLISH>
__device__ __forceinline__ void r16_wave_X(R16Lds &sh, const ModelConsts<double> &c, int p, double TT, double QQ, double VL, double VW, S *out,
                                           bool live, int n, int lane, unsigned long long *stamps_out) {
  using namespace r16;
  RConsts<DevWave> kc;
  make_rconsts(c, kc);
  const int ea = sta_elem(lane), ep = stp_elem(lane);
  const bool wa = live && ea >= 0, wp = live && ep >= 0;
  const int oa = DevWave::iuni(ea >= 0 ? ea : 0), opz = DevWave::iuni(ep >= 0 ? ep : 0);
  const int last = (n - 1) - (((n - 1) & 1) ^ p);  // this wave's last knot (< 0: none)
#ifdef QILQR_STAMPS
  unsigned long long stamp_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp_prev)::"memory");
#endif
  if (p == 1 && n > 1) {  // "step -1": T_1 = T_0 Exp(dt v_0), this wave's first pose and the even wave's first compose
    double DQ, PP, TTn, QQn;
    a_exp<DevWave>(kc, VL, VW, DQ, PP);
    b_compose<DevWave>(kc, TT, QQ, DQ, PP, TTn, QQn);
    TT = TTn;
    QQ = QQn;
    sh.xch[X_T][1][0][lane] = TT;
    sh.xch[X_T][1][1][lane] = QQ;
    r16_flag_post(sh, R16_F_T, 1, lane);
  }
  int seen = -1;  // P's progress as last read
  // One step.  `op`: the knot's operand registers, already requested by the previous step of this wave if `have` (their LDS
  // round trip then lies behind that step's compose); `opn` / `have_n`: the same for this wave's next knot, requested here.
  // The loop calls it with the two register sets alternating (no copies).
  auto step = [&](int i, double (&op)[NOPS], bool have, double (&opn)[NOPS], bool &have_n) -> bool {
    if (!have) {
      if (__builtin_expect(!r16_flag_wait(sh, R16_F_PROD, i + 1, seen, lane), 0)) return false;
#pragma unroll
      for (int r = 0; r < NOPS; ++r) op[r] = sh.ops[i % R16_RING][r][lane];
      seen = r16_flag_read(sh, R16_F_PROD);
    }
    have_n = false;
    S *ok_ = out + (long)i * (9 * TILE2);
    if (wp) ok_[opz] = (S)__builtin_fma(kc.MQ0, QQ, TT);  // [q | t] in one store
    QSTAMP(0);  // X: operand reads (if not requested ahead), pose store
    double TH4, TD;
    b_log<DevWave>(kc, TT, QQ, op, TH4, TD);
    // v_i: the other wave posts it about now (after its control section, half a step ahead of this one's): requested as late as
    // the LDS round trip allows -- a request that comes before the post has to be repeated by polling
    int fv = 0;
    double v[2] = {VL, VW};
    if (i > 0) r16_handoff_request<2>(sh, R16_F_V, X_V, i & 3, fv, v, lane);
    const double RH = a_rho<DevWave>(TH4, TD);
    QSTAMP(1);  // X: Log
    if (i > 0) {
      if (__builtin_expect(!r16_handoff_finish<2>(sh, R16_F_V, i, X_V, i & 3, fv, v, lane), 0)) return false;
    }
    QSTAMP(2);  // X: wait for v_i
    const bool advance = i + 1 < n;  // the reference's step after the last knot is computed and discarded (ilqr.hh:168)
    APre<DevWave> pre;
    a_pre<DevWave>(kc, v[0], v[1], op, pre);
    double VLn = 0.0, VWn = 0.0;
    const double st = a_post<DevWave, true>(kc, pre, TH4, RH, QQ, v[0], op, advance, VLn, VWn);
    if (__builtin_expect(advance, 1)) {  // (the post also tells P that knot i's operand slot is free)
      sh.xch[X_V][(i + 1) & 3][0][lane] = VLn;
      sh.xch[X_V][(i + 1) & 3][1][lane] = VWn;
#ifdef QILQR_DIAG
      if (i != g_r16_stall_knot)  // fault injection (qilqr_debug_set_rollout_stall): this hand-off is never announced
#endif
      r16_flag_post(sh, R16_F_V, i + 1, lane);
    }
    if (wa) ok_[oa] = (S)st;
    if (PUBLISH) r16_publish_stores<PUBLISH>(sh, R16_F_K0 + p, i, last, lane);
    QSTAMP(3);  // X: control, velocity, hand-off, store
    if (i + 2 < n) {
      int ft;
      double t[2];
      r16_handoff_request<2>(sh, R16_F_T, X_T, (i + 1) & 3, ft, t, lane);  // T_{i+1}: posted at the end of the other wave's step i - 1
      double DQ, PP;
      a_exp<DevWave>(kc, VLn, VWn, DQ, PP);
      QSTAMP(4);  // X: Exp
      if (__builtin_expect(!r16_handoff_finish<2>(sh, R16_F_T, i + 1, X_T, (i + 1) & 3, ft, t, lane), 0)) return false;
      QSTAMP(5);  // X: wait for T_{i+1}
      // the operands of this wave's next knot, if P has them (it is normally three knots ahead): requested here, used after
      // the compose
      if (seen >= i + 3) {
#pragma unroll
        for (int r = 0; r < NOPS; ++r) opn[r] = sh.ops[(i + 2) % R16_RING][r][lane];
        seen = r16_flag_read(sh, R16_F_PROD);
        have_n = true;
      }
      double TTn, QQn;
      b_compose<DevWave>(kc, t[0], t[1], DQ, PP, TTn, QQn);
      TT = TTn;
      QQ = QQn;
      sh.xch[X_T][(i + 2) & 3][0][lane] = TT;
      sh.xch[X_T][(i + 2) & 3][1][lane] = QQ;
      r16_flag_post(sh, R16_F_T, i + 2, lane);
      QSTAMP(6);  // X: compose, hand-off
    }
    return true;
  };
  double opA[NOPS], opB[NOPS];
  bool haveA = false, haveB = false;
  for (int i = p; i < n; i += 4) {
    if (!step(i, opA, haveA, opB, haveB)) return;
    if (i + 2 >= n) break;
    if (!step(i + 2, opB, haveB, opA, haveA)) return;
  }
#ifdef QILQR_STAMPS
  if (lane == 0 && stamps_out)
    for (int k = 0; k < 8; ++k) stamps_out[k] = stamp_sum[k];
#endif
}

template <typename S>
__global__ __launch_bounds__(192) void k_rollout16(ModelConsts<double> c, BatchState st, int B, int n, int need_flag) {
#define R16_RETURN return
#include "rollout16_body.inc"
#undef R16_RETURN
}

}  // namespace qilqr
