// cassie_kernels_duo.hip -- gfx950 backend and kernel of the 64-environments-per-wavefront Env.step (cassie_duo_core.h): the lane-per-leg
// set-up of cassie_leg_core.h for two groups of 32 environments, one joint PGS sweep with a lane per environment.
//
//   workgroup b (one wavefront), lane 2e + k:  leg k of environment 64 b + e (group A) AND of environment 64 b + 32 + e (group B)
//   in the sweeps:  even lane 2e = environment 64 b + e, odd lane 2e + 1 = environment 64 b + 32 + e  (both legs each)
//
// LDS (39 168 B per wavefront, four wavefronts per CU -- the budget of the two-lanes kernel, now for twice the environments): per group the
// per-lane slots a substep itself produces and consumes (clock, sum of squared actions, smooth force, link origins, two contact-pair
// descriptors); the third pair and the joint-limit descriptors exist once (a group that needs them is solved before the other group's
// set-up begins).  What the two-lanes kernel kept in LDS for the whole step and touched once -- setState snapshot, qstate, motor commands,
// action -- is read from / written to the HBM record and the action row where it is used (`DuoLds::cld / cst` route the slot numbers).
#ifndef CASSIE_KERNELS_DUO_HIP_
#define CASSIE_KERNELS_DUO_HIP_
#include "cassie_kernels_leg.hip"
#include "cassie_duo_core.h"

namespace cassie {
namespace leg {

#ifndef DUO_WAVES
#define DUO_WAVES 2   // independent wavefronts per workgroup (no barrier, no shared data).  A/B r05, 65 536 envs: 1 -> 1.146 ms, 2 -> 1.125, 4 -> 1.129
#endif
#ifndef DUO_STAGGER
#define DUO_STAGGER 0
#endif
#ifndef DUO_WS_PAD
#define DUO_WS_PAD 0     // slots of padding between the workspaces of two wavefronts (A/B: an odd number of 512-byte slots per wavefront)
#endif
#ifndef DUO_WS_LD_AUX
#define DUO_WS_LD_AUX 0  // cache-policy bits of the workspace loads / stores (A/B: 2 = nt)
#endif
#ifndef DUO_WS_ST_AUX
#define DUO_WS_ST_AUX 0
#endif
#if DUO_WAVES == 1
#define DUO_LANE ((int)threadIdx.x)
#else
#define DUO_LANE ((int)threadIdx.x & 63)
#endif
__device__ const double duo_zero_action[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};   // the action row of a step without actions
struct DuoShared {
  double cold[2][20][64];    // per group: clock, a2, tau_b 3, tau_l 5, link origins x 5, z 5 (the pelvis origin is the constant 0)
  double pr[2][2][4][64];    // per group: contact pairs 0, 1
  double pr2[4][64];         // third pair (eight-row path only)
  double lm[4][3][64];       // joint limits (eight-row path only)
  int pdepth[2][2][64];
  int pdepth2[64];
  int lmj[4][64];
};
static_assert(sizeof(DuoShared) == 39168, "LDS budget of four wavefronts per CU");

struct DevDuoB : DevB {
#ifndef DUO_SPLIT_TAIL
#define DUO_SPLIT_TAIL 1
#endif
  // sub_setup: a group on its feet does not build its two empty row slots and warm-starts over six.  r05 had this off for this kernel (the branch cost 75 more
  // spills with the eight-row pair solve inline behind the set-up: 1.01 -> 1.08 ms); r06: the eight-row groups leave for the joint sweep like the others, the
  // set-up's two branches end in their own stores to the workspace and never meet again: 304 B of scratch instead of 264, **1.018 -> 0.990 ms** per 65 536-env
  // step (tools/ab_bench.py, both orders), bit-identical.
  static constexpr bool SPLIT_TAIL = DUO_SPLIT_TAIL != 0;
  // per-wavefront workspace in global memory (Duo::W_*): [slot][lane], the lane's pointer is the base of its column
  // Buffer addressing: one resource descriptor per wavefront (SGPRs), the lane's byte offset in ONE VGPR, the slot as the scalar offset
  // of the instruction -- so a slot costs an s_mov, not a 64-bit per-lane pointer (with plain pointers the compiler materialises one
  // pointer per slot beyond the 4 KB immediate range, ~150 of them, and spills them: first build of this path).
  struct W { __amdgpu_buffer_rsrc_t r; unsigned voff; };
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  // slots 2p and 2p + 1 of a lane are adjacent: [p][lane][2] -- a pair of slots is ONE sixteen-byte access per lane (1 KB per wavefront)
  static LEG_FN constexpr int wofs(int slot) { return (slot >> 1) * 1024 + (slot & 1) * 8; }
  static LEG_FN double wld(W ws, int slot) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(ws.r, ws.voff, wofs(slot), DUO_WS_LD_AUX);
    return __hiloint2double((int)v.y, (int)v.x);
  }
  static LEG_FN void wst(W ws, int slot, double v) {
    u32x2 w; w.x = (unsigned)__double2loint(v); w.y = (unsigned)__double2hiint(v);
    __builtin_amdgcn_raw_buffer_store_b64(w, ws.r, ws.voff, wofs(slot), DUO_WS_ST_AUX);
  }
  static LEG_FN void wld2(W ws, int slot, double& a, double& b) {   // slot even
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ws.r, ws.voff, wofs(slot), DUO_WS_LD_AUX);
    a = __hiloint2double((int)v.y, (int)v.x); b = __hiloint2double((int)v.w, (int)v.z);
  }
  static LEG_FN void wst2(W ws, int slot, double a, double b) {
    u32x4 w; w.x = (unsigned)__double2loint(a); w.y = (unsigned)__double2hiint(a); w.z = (unsigned)__double2loint(b); w.w = (unsigned)__double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(w, ws.r, ws.voff, wofs(slot), DUO_WS_ST_AUX);
  }
#ifdef DUO_VIEW_EXPERIMENT
  // the view of joint lane 2e + k on column 2e + X of group k's block (group stride in slots; see joint_solve_view)
  static LEG_FN W wview(W ws, int group_slots, int X) {
    W v; v.r = ws.r;
    const unsigned lane = ws.voff >> 4;
    v.voff = ((lane & ~1u) + (unsigned)X) * 16u + (lane & 1u) * (unsigned)(group_slots / 2) * 1024u;
    return v;
  }
  template <int N> static LEG_FN void wput_if(W ws, int first, double (&t)[N], bool m) {
    if (m) { for (int i = 0; i < N; i++) wst(ws, first + i, t[i]); }
  }
  static LEG_FN void wst_if(W ws, int slot, double v, bool m) { if (m) wst(ws, slot, v); }
#endif
  struct Lds {
    DuoShared* sh;
    int g;
    double* rec;
    const double* act;
    bool has_act, snap;
    int lo, ao;
#ifdef CASSIE_PHASE_TIMING   // profiling builds (tools/phase_profile.py duo): shader cycles per phase of this wavefront; the time up to mark(k) goes to bucket k
    unsigned long long t_last, acc[16];
    LEG_FN void mark(int k) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long n = __builtin_readcyclecounter();
#ifdef DUO_GLUE_SPLIT   // experiment: the glue in pieces (marks 16..21 -> buckets 1..6), everything else in bucket 7
      acc[k >= 16 ? k - 15 : k == 0 ? 0 : 7] += n - t_last; t_last = n;
#else
      acc[k < 16 ? k : 0] += n - t_last; t_last = n;
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
#else
    LEG_FN void mark(int) {}
#endif
    template <class IoT> LEG_FN void select(int group, const IoT& io) { g = group; rec = io.rec; act = io.act; has_act = io.has_act; }
    LEG_FN void snapshot(bool on) { snap = on; }
    // slot of the group's cold block, or -1: not in LDS
    static LEG_FN constexpr int slot(int i) {
      return i == 24 ? 0 : i == 28 ? 1 : (i >= 29 && i < 37) ? i - 27 : (i >= 38 && i < 43) ? i - 28 : (i >= 44 && i < 49) ? i - 29 : -1;
    }
    LEG_FN double cld(int i) const {
      const int l = DUO_LANE;
      if (i < 8) return rec[ES_KQ + (i < 3 ? i : lo + (i - 3))];
      if (i < 16) return rec[ES_KV + (i - 8 < 3 ? i - 8 : lo + (i - 11))];
      if (i < 21) return rec[ES_QSTATE + lo + (i - 16)];
      if (i < 24) return rec[ES_CTRL + ao + (i - 21)];
      if (i >= 25 && i < 28) return (has_act ? act : duo_zero_action)[ao + (i - 25)];   // (one unconditional load: three of them go out in one batch)
      if (i == 37 || i == 43) return 0.0;   // origin of the pelvis link relative to the pelvis origin
      return sh->cold[g][slot(i)][l];
    }
    LEG_FN void cst(int i, double v, bool m) {
      const int l = DUO_LANE;
      if (i < 16) {
        if (snap && m) { if (i < 8) rec[ES_KQ + (i < 3 ? i : lo + (i - 3))] = v; else rec[ES_KV + (i - 8 < 3 ? i - 8 : lo + (i - 11))] = v; }
      } else if (i < 21) { if (m) rec[ES_QSTATE + lo + (i - 16)] = v; }
      else if (i < 24) { if (m) rec[ES_CTRL + ao + (i - 21)] = v; }
      else if (i >= 25 && i < 28) {}
      else if (i == 37 || i == 43) {}
      else { if (m) sh->cold[g][slot(i)][l] = v; }
    }
    LEG_FN void st_pair(int s, double px, double pz, double dist, double invw, int depth, bool m) {
      if (m) {
        const int l = DUO_LANE;
        double* p = s < 2 ? &sh->pr[g][s][0][l] : &sh->pr2[0][l];
        p[0] = px; p[64] = pz; p[128] = dist; p[192] = invw;
        *(s < 2 ? &sh->pdepth[g][s][l] : &sh->pdepth2[l]) = depth;
      }
    }
    LEG_FN void ld_pair(int s, double& px, double& pz, double& dist, double& invw, int& depth) const {
      const int l = DUO_LANE;
      if (s < 2) { px = sh->pr[g][s][0][l]; pz = sh->pr[g][s][1][l]; dist = sh->pr[g][s][2][l]; invw = sh->pr[g][s][3][l]; depth = sh->pdepth[g][s][l]; }
      else { px = sh->pr2[0][l]; pz = sh->pr2[1][l]; dist = sh->pr2[2][l]; invw = sh->pr2[3][l]; depth = sh->pdepth2[l]; }
    }
    LEG_FN void st_lim(int slot_, double pos, double sgn, double invw, int j, bool m) {
      if (m) {
        const int l = DUO_LANE;
        sh->lm[slot_][0][l] = pos; sh->lm[slot_][1][l] = sgn; sh->lm[slot_][2][l] = invw; sh->lmj[slot_][l] = j;
      }
    }
    LEG_FN void ld_lim(int s, double& pos, double& sgn, double& invw, int& j) const {
      const int l = DUO_LANE;
      pos = sh->lm[s][0][l]; sgn = sh->lm[s][1][l]; invw = sh->lm[s][2][l]; j = sh->lmj[s][l];
    }
  };
};

typedef Duo<DevDuoB> DDuo;
static_assert(DDuo::C::C_TIME == 24 && DDuo::C::C_A2 == 28 && DDuo::C::C_TAUB == 29 && DDuo::C::C_OX == 37 && DDuo::C::C_OZ == 43 && DDuo::C::C_N == 49 &&
              DDuo::C::C_KQ == 0 && DDuo::C::C_KV == 8 && DDuo::C::C_QST == 16 && DDuo::C::C_CTRL == 21 && DDuo::C::C_ACT == 25, "DuoLds routes these slot numbers");

// MODE: 0 PD, 1 torque, 2 motor commands from the state record.  pending[env] as env_step_leg_kernel.  workspace: W_N slots x 64 lanes per
// wavefront SLOT (launch::duo_workspace_bytes; contents only live inside one task of one launch).
constexpr size_t duo_workspace_doubles_per_wave = (size_t)(DDuo::W_N + DUO_WS_PAD) * 64;
// THE WORKSPACE IS A PROPERTY OF THE CHIP, NOT OF THE BATCH (r06).  One wavefront of this kernel owns a SIMD (512 registers), so at most
// 4 x CUs wavefronts exist at any time, however many the launch has.  A batch of up to DuoSlots::DIRECT_MAX tasks (one round of an MI355X,
// the headline's 65 536 envs) indexes the workspace by its task number, as r05 did.  A larger batch CLAIMS a slot per wavefront from a table
// of `mask + 1` busy words (a power of two >= 2 x the chip's SIMDs, one eighth of it per XCD): the first probe is a hash of the wavefront's physical place
// (XCC, SE, CU, SIMD from HW_REG_XCC_ID / HW_REG_HW_ID) -- two resident wavefronts of this kernel cannot share a SIMD, so the probe finds its
// word free and the same SIMD comes back to the same 176 KB (of which a batch on its feet touches 139), launch after launch: 180 MB touched on an MI355X for any batch (360 MB allocated: two words per SIMD), Infinity-Cache
// resident (r05 indexed by the task: 1.14 GB at 524 288 envs).  The place is only a HINT: the claim is an atomic compare-and-swap with linear
// probing, so a collision (another hash layout, a future part, the test mode that zeroes the hint) costs probes (STAT_WS_PROBES), never a
// result.  The hardware's dispatcher stays the task queue: a wavefront that lost the arbitration for the fabric (-7..+10 % lifetime spread,
// profiles/r05_phase_duo.txt) simply frees its SIMD later, and the next workgroup goes where a SIMD is free.
// (Built and measured first as a persistent grid pulling tasks from a counter: the task loop around the body cost 68 B more scratch per lane
// and +1.5 % at every batch size, profiles/r06_size_sweep.jsonl.)
struct DuoSlots {
  static constexpr int DIRECT_MAX = 1024;   // tasks up to which the workspace is indexed by the task (host: cassie_cabi.hip sizes it)
  unsigned* busy;      // null: slot = task
  unsigned mask;       // table size - 1
  unsigned flat_hint;  // tests: every wavefront starts probing at word 0
  __device__ __forceinline__ int claim(int lane, int task, unsigned long long* stats) const {
    if (!busy) return task;
    unsigned h = 0;
    if (lane == 0) {
      // HW_REG_HW_ID (4): WAVE_ID [3:0], SIMD_ID [5:4], PIPE_ID [7:6], CU_ID [11:8], SH_ID [12], SE_ID [15:13]; HW_REG_XCC_ID (20): XCC_ID [3:0]
      // (s_getreg operand: size - 1 << 11 | offset << 6 | register).  MI355X census (profiles/tools/hwid_census.hip, profiles/r06_hwid_census.txt):
      // XCC 0..7, SE 0..3, SH 0, CU 0..8, SIMD 0..3 -- 1152 places, 1024 of them active; key = xcc:3 | se:2 | cu:4 | simd:2 is injective there
      // (2048 words).  Bits a larger part might use (SE bit 2, SH, XCC bit 3) are folded in: they can only cost probes.
      const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
      const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
      const unsigned simd = (hw >> 4) & 3u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
      // The table is PARTITIONED BY XCD (xcc in the top three bits of the word index, probing wraps inside the partition): a slot is only ever
      // re-used by wavefronts of one XCD, i.e. through one L2 -- the per-XCD L2s are not coherent with each other inside a launch, and a slot
      // that moved between XCDs would need its old owner's dirty lines written back first (an agent-scope release = buffer_wbl2 of the whole
      // L2: measured, 5 % of the launch at 131 072 envs).
      const unsigned part = (mask + 1u) >> 3, pm = part - 1u;
      const unsigned base = (xcc & 7u) * part;
      unsigned k = flat_hint ? 0u : ((se & 3u) << 6 | cu << 2 | simd) + 37u * (sh + 2u * (se >> 2) + 4u * (xcc >> 3));
      unsigned probes = 0;
      while (atomicCAS(busy + base + (k & pm), 0u, 1u) != 0u) { k++; probes++; }
      h = base + (k & pm);
      if (probes && stats) atomicAdd(stats + STAT_WS_PROBES, (unsigned long long)probes);
    }
    return __builtin_amdgcn_readfirstlane((int)h);
  }
  __device__ __forceinline__ void release(int lane, int slot) const {
    // every store of this task to the workspace has reached the L2 (vmcnt counts stores until they are acknowledged) before the word is freed;
    // the next owner is a wavefront of the same XCD (see claim), so no write-back of the L2 is needed: a relaxed store
    if (busy) {
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (gfx9 encoding: vmcnt in bits 3:0 and 15:14, expcnt 6:4 and lgkmcnt 11:8 left at their maxima)
      if (lane == 0) __hip_atomic_store(busy + slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
};

template <int MODE>
__global__ void __launch_bounds__(64 * DUO_WAVES, 1) env_step_duo_kernel(VecParams p, int* pending, double* workspace, DuoSlots sl) {
#if DUO_WAVES == 1
  __shared__ DuoShared sh;
  const int lane = threadIdx.x;
  const int wave_id = blockIdx.x;
#else
  __shared__ DuoShared shs[DUO_WAVES];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);   // wave-uniform: LDS addressing stays on a scalar base
  DuoShared& sh = shs[wv];
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * DUO_WAVES + wv;
#endif
  EnvCfg cfg;
  cfg.n_sub = p.n_sub; cfg.flags = p.flags; cfg.env_kind = p.env_kind; cfg.auto_reset = p.auto_reset; cfg.adim = p.adim;
  cfg.want_obs = p.obs != nullptr; cfg.traj_qpos = p.traj_qpos; cfg.traj_tmax = p.traj_tmax; cfg.traj_n = p.traj_n;
  bool valid[2];
  size_t e[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int env = wave_id * 64 + g * 32 + (lane >> 1);
    valid[g] = env < p.n_envs;
    e[g] = valid[g] ? (size_t)env : 0;
  }
  // the group's per-lane pointers, rebuilt where a phase needs them (held for the whole kernel they are 24 registers the allocator spills)
  auto io_of = [&](int g) {
    const int env = wave_id * 64 + g * 32 + (lane >> 1);
    const size_t eg = env < p.n_envs ? (size_t)env : 0;
    DDuo::Io io;
    io.rec = p.state + eg * ENV_STRIDE;
    io.has_act = p.actions != nullptr;
    io.act = const_cast<double*>(p.actions) + (io.has_act ? eg * p.adim : 0);
    io.obs = p.obs + (cfg.want_obs ? eg * 26 : 0);
    io.has_tobs = p.terminal_obs != nullptr;
    io.tobs = p.terminal_obs + (io.has_tobs ? eg * 26 : 0);
    io.rew = p.reward + (cfg.want_obs ? eg : 0);
    io.done = p.done + (cfg.want_obs ? eg : 0);
    return io;
  };
  DevDuoB::Lds lds;
  lds.sh = &sh; lds.g = 0; lds.rec = p.state; lds.act = p.actions; lds.has_act = false; lds.snap = true;
  lds.lo = (lane & 1) * 5 + 3; lds.ao = (lane & 1) * 3;
#if DUO_STAGGER > 0
  // Stagger: every wavefront runs the same phases for the same time, so all 128 wavefronts of an XCD hit its L2 with their hand-over bursts at
  // once (phase clocks, r05: 16 % of a wavefront's time in the hand-over, at ~60 cycles per 512-byte access = the L2's bandwidth shared by 128).
  // Eight start classes DUO_STAGGER x 3.4 us apart keep the bursts of most wavefronts apart for the whole launch.
  {
#if DUO_WAVES == 1
    const int cls = (blockIdx.x >> 3) & 7;
#else
    const int cls = (((blockIdx.x >> 3) & 3) * 2 + (wv & 1)) & 7;
#endif
    for (int i = 0; i < cls * DUO_STAGGER; i++) __builtin_amdgcn_s_sleep(127);
  }
#endif
  DDuo::Out o[2];
  DevDuoB::W ws;   // raw buffer over this wavefront's W_N x 512 bytes (word 3: 32-bit data format, gfx9 encoding)
  const int slot = sl.claim(lane, wave_id, p.stats);
  ws.r = __builtin_amdgcn_make_buffer_rsrc(workspace + (size_t)slot * duo_workspace_doubles_per_wave, 0, DDuo::W_N * 512, 0x00020000);
  ws.voff = (unsigned)lane * 16u;
#ifdef CASSIE_PHASE_TIMING
  for (int i = 0; i < 16; i++) lds.acc[i] = 0;
  lds.t_last = __builtin_readcyclecounter();
#endif
  DDuo::env_step2<MODE>(cfg, lds, ws, io_of, valid, o);
  sl.release(lane, slot);   // (the write-back at the end of env_step2 has consumed its loads from the workspace: they fed its stores to the records)
#ifdef CASSIE_PHASE_TIMING
  lds.mark(0);
  if (lane == 0 && p.phase) for (int i = 0; i < 16; i++) atomicAdd(p.phase + i, lds.acc[i]);
#endif
#pragma unroll
  for (int g = 0; g < 2; g++) {
    if (valid[g] && (lane & 1) == 0) {
      pending[e[g]] = o[g].pend;
      if (p.stats) {
        if (o[g].pend > 0) atomicAdd(p.stats + STAT_CLEANUP_SUBSTEPS, (unsigned long long)o[g].pend);
        if (o[g].bad) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
      }
    }
  }
}


#ifdef CASSIE_LEG_HF
// ---------------------------------------------------------------- height-field instantiation (tu_duo_hf.hip)
// One more per-lane word per contact pair (the x component of the local terrain normal).  LDS stays at four wavefronts per CU: pair 0's normal
// takes the slot of the sum of squared actions (which this backend keeps in two registers instead), pair 1's and the third pair's get three new
// slots: 39 168 + 1536 = 40 704 B.
struct DuoSharedHF : DuoShared {
  double nrm1[2][64];
  double nrm2[64];
};
static_assert(sizeof(DuoSharedHF) == 40704, "LDS budget of four wavefronts per CU");

struct DevDuoBHF : DevDuoB {
  struct Lds : DevDuoB::Lds {
    DuoSharedHF* shf;
    double a2[2];
    LEG_FN double cld(int i) const { return i == 28 ? (g == 0 ? a2[0] : a2[1]) : DevDuoB::Lds::cld(i); }
    LEG_FN void cst(int i, double v, bool m) {
      if (i == 28) { if (m) { if (g == 0) a2[0] = v; else a2[1] = v; } }
      else DevDuoB::Lds::cst(i, v, m);
    }
    LEG_FN void st_nrm(int slot, double nx, bool m) {
      if (m) { const int l = DUO_LANE; *(slot == 0 ? &shf->cold[g][1][l] : slot == 1 ? &shf->nrm1[g][l] : &shf->nrm2[l]) = nx; }
    }
    LEG_FN double ld_nrm(int s) const { const int l = DUO_LANE; return s == 0 ? shf->cold[g][1][l] : s == 1 ? shf->nrm1[g][l] : shf->nrm2[l]; }
  };
  static LEG_FN void hf_sphere(const Terrain& t, double wx, double wy, double wz, double radius, double& dist, double& nx, double& nz) {
    cassie::terrain_sphere(t, wx, wy, wz, radius, dist, nx, nz);
  }
};
typedef Duo<DevDuoBHF> DDuoHF;
static_assert(DDuoHF::W_N == DDuo::W_N, "one workspace size for both instantiations");

template <int MODE>
__global__ void __launch_bounds__(64 * DUO_WAVES, 1) env_step_duo_hf_kernel(VecParams p, int* pending, double* workspace, DuoSlots sl) {
#if DUO_WAVES == 1
  __shared__ DuoSharedHF sh;
  const int lane = threadIdx.x;
  const int wave_id = blockIdx.x;
#else
  __shared__ DuoSharedHF shs[DUO_WAVES];
  const int wv = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  DuoSharedHF& sh = shs[wv];
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * DUO_WAVES + wv;
#endif
  EnvCfg cfg;
  cfg.n_sub = p.n_sub; cfg.flags = p.flags; cfg.env_kind = p.env_kind; cfg.auto_reset = p.auto_reset; cfg.adim = p.adim;
  cfg.want_obs = p.obs != nullptr; cfg.traj_qpos = p.traj_qpos; cfg.traj_tmax = p.traj_tmax; cfg.traj_n = p.traj_n;
  bool valid[2];
  size_t e[2];
#pragma unroll
  for (int g = 0; g < 2; g++) {
    const int env = wave_id * 64 + g * 32 + (lane >> 1);
    valid[g] = env < p.n_envs;
    e[g] = valid[g] ? (size_t)env : 0;
  }
  auto io_of = [&](int g) {
    const int env = wave_id * 64 + g * 32 + (lane >> 1);
    const size_t eg = env < p.n_envs ? (size_t)env : 0;
    DDuoHF::Io io;
    io.rec = p.state + eg * ENV_STRIDE;
    io.has_act = p.actions != nullptr;
    io.act = const_cast<double*>(p.actions) + (io.has_act ? eg * p.adim : 0);
    io.obs = p.obs + (cfg.want_obs ? eg * 26 : 0);
    io.has_tobs = p.terminal_obs != nullptr;
    io.tobs = p.terminal_obs + (io.has_tobs ? eg * 26 : 0);
    io.rew = p.reward + (cfg.want_obs ? eg : 0);
    io.done = p.done + (cfg.want_obs ? eg : 0);
    return io;
  };
  DevDuoBHF::Lds lds;
  lds.sh = &sh; lds.shf = &sh; lds.g = 0; lds.rec = p.state; lds.act = p.actions; lds.has_act = false; lds.snap = true;
  lds.lo = (lane & 1) * 5 + 3; lds.ao = (lane & 1) * 3;
  lds.a2[0] = 0.0; lds.a2[1] = 0.0;
  DevDuoBHF::W ws;
  const int slot = sl.claim(lane, wave_id, p.stats);   // (as env_step_duo_kernel)
  ws.r = __builtin_amdgcn_make_buffer_rsrc(workspace + (size_t)slot * duo_workspace_doubles_per_wave, 0, DDuoHF::W_N * 512, 0x00020000);
  ws.voff = (unsigned)lane * 16u;
  DDuoHF::Out o[2];
  DDuoHF::env_step2<MODE, true>(cfg, lds, ws, io_of, valid, o, &p.hf);
  sl.release(lane, slot);
#pragma unroll
  for (int g = 0; g < 2; g++) {
    if (valid[g] && (lane & 1) == 0) {
      pending[e[g]] = o[g].pend;
      if (p.stats) {
        if (o[g].pend > 0) atomicAdd(p.stats + STAT_CLEANUP_SUBSTEPS, (unsigned long long)o[g].pend);
        if (o[g].bad) atomicAdd(p.stats + STAT_NONFINITE, 1ull);
      }
    }
  }
}
#endif

}  // namespace leg
}  // namespace cassie
#endif
