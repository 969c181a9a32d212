// lane_types.h -- CHECKER / CPU-BASELINE INFRASTRUCTURE: the lane emulation shared by the CPU instantiations of the lane-per-leg
// kernel sources (leg_host.cpp: cassie_leg_core.h, Cassie2d; leg3d_host.cpp: cassie3d_leg_core.h, Cassie3d).  Lane values are GCC
// vector-extension types, NL lanes = NL / 2 environments; HostOps holds the backend primitives both cores use.
#ifndef LEG_HOST_LANE_TYPES_H_
#define LEG_HOST_LANE_TYPES_H_
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef LEG_HOST_LANES
#define LEG_HOST_LANES 2
#endif

namespace {


constexpr int NL = LEG_HOST_LANES;   // lanes per group: lane l = leg (l & 1) of the group's environment (l >> 1)
static_assert(NL >= 2 && NL % 2 == 0, "two lanes per environment");
#define LANES for (int l = 0; l < NL; l++)
typedef long long i64;   // every lane type is 64 bits wide, so that one AVX-512 register holds eight lanes of any of them

// Op counting (tools/count_flops.py): every arithmetic operation on a lane value adds 1 per COUNTED lane (a*b+c is written as a
// multiply and an add in the core: 2); divisions, square roots and reciprocals count 1, sincos 2, exp 1; comparisons and selects 0.
// Inside a Gauss-Seidel step only the owner leg's lane is counted (the other lane executes the same instructions on values
// that are thrown away); everywhere else both lanes are.  Compiled out of the timing build.
#ifdef LEG_HOST_FAST
inline void ops(int = 1) {}
#else
double g_ops = 0.0;
bool g_cnt[NL];
struct CntInit { CntInit() { LANES g_cnt[l] = true; } } g_cnt_init;
inline void ops(int k = 1) { int c = 0; LANES c += (int)g_cnt[l]; g_ops += k * c; }
#endif

// Lane values are GCC vector-extension types (NL x 64 bit): with NL = 8 and -march=native every operation below is one AVX-512
// instruction; with NL = 2 one SSE2 instruction.  Masks are what vector comparisons give: all-ones / zero per lane.
typedef double vdn __attribute__((vector_size(NL * 8)));
typedef i64 vin __attribute__((vector_size(NL * 8)));
struct VM {
  vin v;
  VM() {}
  VM(bool b) { v = vin{} - (i64)b; }
  VM(vin x) : v(x) {}
};
struct VI {
  vin v;
  VI() {}
  VI(int a) { v = vin{} + (i64)a; }
  VI(vin x) : v(x) {}
};
struct VD {
  vdn v;
  VD() {}
  VD(double a) { v = vdn{} + a; }
  VD(vdn x) : v(x) {}
};
#define VD_BIN(op) inline VD operator op(const VD& a, const VD& b) { ops(); return VD(a.v op b.v); }
VD_BIN(+) VD_BIN(-) VD_BIN(*) VD_BIN(/)
inline VD operator-(const VD& a) { return VD(-a.v); }
inline VD& operator+=(VD& a, const VD& b) { a = a + b; return a; }
#define VD_CMP(op) inline VM operator op(const VD& a, const VD& b) { return VM((vin)(a.v op b.v)); }
VD_CMP(<) VD_CMP(>) VD_CMP(<=) VD_CMP(>=) VD_CMP(==)
#define VI_BIN(op) inline VI operator op(const VI& a, const VI& b) { return VI(a.v op b.v); }
VI_BIN(+) VI_BIN(-) VI_BIN(*)
#define VI_CMP(op) inline VM operator op(const VI& a, const VI& b) { return VM((vin)(a.v op b.v)); }
VI_CMP(<) VI_CMP(>) VI_CMP(<=) VI_CMP(>=) VI_CMP(==) VI_CMP(!=)
inline VM operator&(const VM& a, const VM& b) { return VM(a.v & b.v); }
inline VM operator|(const VM& a, const VM& b) { return VM(a.v | b.v); }
inline VM operator!(const VM& a) { return VM(~a.v); }
inline vin lane_swap_idx() { vin r; LANES r[l] = l ^ 1; return r; }
template <int W> inline vin lane_bcast_idx() { vin r; LANES r[l] = (l & ~1) | W; return r; }

struct HostOps {
  static constexpr bool SPLIT_TAIL = true;   // (cassie_leg_core.h: sub_setup; the duo emulation says false, as the device backends do)
  typedef VD D;
  typedef VI I;
  typedef VM M;
#ifdef LEG_HOST_FAST
  struct OwnerScope { OwnerScope(const VM&) {} };
#else
  struct OwnerScope {
    bool old[NL];
    OwnerScope(VM owner) { LANES { old[l] = g_cnt[l]; g_cnt[l] = owner.v[l] != 0; } }
    ~OwnerScope() { LANES g_cnt[l] = old[l]; }
  };
#endif
  struct P { double* p[NL]; };
  struct P8 { uint8_t* p[NL]; };
  static VI leg() { VI r; LANES r.v[l] = l & 1; return r; }
  static VI swapi(VI x) { return VI(__builtin_shuffle(x.v, lane_swap_idx())); }
  static VI opq(VI x) { return x; }
  static void fence() {}
  static int zs() { return 0; }
  static VD sel(VM m, VD a, VD b) { return VD(m.v ? a.v : b.v); }
  static VI seli(VM m, VI a, VI b) { return VI(m.v ? a.v : b.v); }
  static VD swap(VD x) { return VD(__builtin_shuffle(x.v, lane_swap_idx())); }
  template <int W> static VD pair_bcast(VD x) { return VD(__builtin_shuffle(x.v, lane_bcast_idx<W>())); }
  static VM swapm(VM x) { return VM(__builtin_shuffle(x.v, lane_swap_idx())); }
  static bool any(VM m) { i64 a = 0; LANES a |= m.v[l]; return a != 0; }
  static VD ldc(const double* t, VI i) { VD r; LANES r.v[l] = t[i.v[l]]; return r; }
  static VD ldg(const double* t, VI i) { return ldc(t, i); }
  static VI toI(VM m) { return VI(m.v & 1); }
  static VD toD(VI i) { VD r; LANES r.v[l] = (double)i.v[l]; return r; }
  static VI toint(VD x) { VI r; LANES r.v[l] = (int)x.v[l]; return r; }
  static void sincos(VD x, VD& s, VD& c) { ops(2); LANES { s.v[l] = std::sin(x.v[l]); c.v[l] = std::cos(x.v[l]); } }
  static VD sqrt(VD x) { ops(); VD r; LANES r.v[l] = std::sqrt(x.v[l]); return r; }
  static VD rcp(VD x) { ops(); return VD(1.0 / x.v); }
  static VD fma(VD a, VD b, VD c) { ops(2); VD r; LANES r.v[l] = __builtin_fma(a.v[l], b.v[l], c.v[l]); return r; }
  static VD fabs(VD x) { return VD((vdn)((vin)x.v & (vin{} + (i64)0x7fffffffffffffffLL))); }
  static VD fmax(VD a, VD b) { return VD(a.v > b.v ? a.v : b.v); }   // operands are never NaN here
  static VD exp(VD x) { ops(); VD r; LANES r.v[l] = std::exp(x.v[l]); return r; }
  static VD fmod(VD a, double b) { VD r; LANES r.v[l] = std::fmod(a.v[l], b); return r; }
  static VD copysign(VD a, VD b) {
    const vin sign = vin{} + (i64)0x8000000000000000ULL;
    return VD((vdn)(((vin)a.v & ~sign) | ((vin)b.v & sign)));
  }
  static VD pld(P p, VI off) { VD r; LANES r.v[l] = p.p[l][off.v[l]]; return r; }
  static void pst(P p, VI off, VD v, VM m) { LANES if (m.v[l]) p.p[l][off.v[l]] = v.v[l]; }
  static void pst8(P8 p, VM v, VM m) { LANES if (m.v[l]) *p.p[l] = v.v[l] ? 1 : 0; }
};

}  // namespace
#endif
