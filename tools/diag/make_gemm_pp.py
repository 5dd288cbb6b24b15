"""EXPERIMENT (round 3, measured and NOT adopted): generate a PAIRED form of the 256 x 256 GEMM from gemm256.hip's text.

256 x 128 tile, four waves (2 x 2) with gemm256's 128 x 64 wave tile and its whole epilogue, operands staged 32 K elements at a time
through a three-stage LDS ring (72 KiB: two workgroups per CU, one computing while the other stores and refills), one barrier per
stage, 64-B LDS rows with the chunk swizzle c ^ {0, 2, 3, 1}[(row >> 2) & 3] (conflict-free for ds_read_b128).

Result on an MI355X (tools/bench_gemm256.py with the kernel routed behind VLA_GEMM_TILE=7; bit-identical to the other kernels on 27
shape / epilogue cases): 0.62 - 0.88 x gemm256's rate on every shape of the step (LLM down 145 vs 90 us, gate/up live 240 vs 194,
d->dh 112 vs 99, ViT qkv 82 vs 63, o 33 vs 26) - the simple ring leaves each wave ~45 % MFMA duty (six LDS-DMA issues and twelve
fragment reads per 32 MFMAs, a barrier per stage), and two such waves per SIMD do not interleave into the 96 - 98 % that gemm256's
two-phase loop holds; hiding the epilogue does not pay for that.  Kept as the record of the attempt:

    python tools/diag/make_gemm_pp.py > vla_adapter_amd/csrc/gemm_pp.hip      # + Makefile SRCS, a declaration of vla_gemm_pp_launch in
                                                                              #   gemm_params.h and a route in gemm.hip
"""
import sys
import os
src=open(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'vla_adapter_amd', 'csrc', 'gemm256.hip')).read()
s=src
def rep(old,new,cnt=1):
    global s
    assert s.count(old)==cnt,(s.count(old),old[:90])
    s=s.replace(old,new)
def cut(a,b,new,inclusive_b=False):
    """replace from the start of marker a up to (not including) marker b"""
    global s
    i=s.index(a); j=s.index(b,i)
    if inclusive_b: j+=len(b)
    s=s[:i]+new+s[j:]

# ---- header comment
i=s.index('#include <type_traits>')
s='''// bf16 NT GEMM, PAIRED form: 256 x 128 tile, four waves as 2 (M) x 2 (N), wave tile 128 x 64 (the accumulator layout and the whole
// epilogue of gemm256.hip), operands staged 32 K elements at a time through a three-stage LDS ring (24 KiB per stage, 72 KiB per
// workgroup) - so that TWO workgroups share a CU and one computes while the other stores its tile and refills its ring.
// gemm256.hip's single workgroup per CU runs its K loop at the MFMA pipe's cycle count but pays 13 - 20k cycles around every
// tile's 30k-cycle K loop on the K ~ 1000 products of the step (stamps: DESIGN section 4); this form trades a simpler K loop (one
// barrier per 32-deep stage, six LDS-DMAs and twelve fragment reads per 32 MFMAs and wave) for that overlap.
// LDS image of a stage: A rows [0, 256) then B rows [0, 128), 64 B each, 16-B chunk c of row r at slot c ^ {0, 2, 3, 1}[(r >> 2) & 3]:
// ds_read_b128 serves lanes in the groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32, and a fragment read (lane: row l & 15,
// chunk l >> 4) then touches each of the sixteen 16-B bank quads once per group.
// Generated from gemm256.hip's text by tools/diag/make_gemm_pp.py at the time of writing; maintained by hand since.
'''+s[i:]

# ---- constants
rep('''constexpr int BK = 64;
constexpr int HT = 16384;            // bytes per half-tile (128 rows x 64 k x 2 B)
constexpr int LDS_BYTES = 8 * HT;    // 128 KiB of operands: one workgroup per CU''','''constexpr int BK = 32;               // K elements per stage
constexpr int A_ST = 256 * 64;       // bytes of A per stage (256 rows x 64 B)
constexpr int ST = A_ST + 128 * 64;  // bytes per stage
constexpr int LDS_BYTES = 3 * ST;    // 72 KiB: two workgroups per CU''')
# quadrant macro: one 32-deep k-step
cut('#define VLA_MMA_QUADRANT(Q, FB, FA)','typedef int v8i_f8','''#define VLA_MMA_QUADRANT(Q, FB, FA)                                                                       \\
  do {                                                                                                    \\
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                      \\
      _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                    \\
        Q[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(FB[ni], FA[mi], Q[ni][mi], 0, 0, 0);          \\
  } while (0)

''')
# kernel head
rep('''__global__ __launch_bounds__(512) void gemm256_kernel(GemmP p) {''','''__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(GemmP p) {
  static_assert(!F8 && !SSQ && !RN, "the paired form takes bf16 operands and has no RMSNorm fold");''')
rep('''  constexpr bool PRE = EPI != 2;      // next tile's K-tile 0 in flight during the epilogue (SwiGLU backward stages wider rows)''','''  constexpr bool PRE = false;         // (the ring is the epilogue's staging area: the next tile's first stage goes out behind the epilogue)''')
rep('''  const int wr = wid >> 2, wc = wid & 3;''','''  const int wr = wid >> 1, wc = wid & 1;''')
# setup(): lane mapping + offsets
rep('''  unsigned oa[2][2], ob[2][2];''','''  unsigned oa[4], ob[2];''')
rep('''    const int kc = ((sl & 7) ^ ((sl >> 3) & 7)) * 16;  // (bytes) across the K loop); LDS chunk lane&7 of row r holds global chunk (lane&7)^(r&7)
    const int lrow = sl >> 3;''','''    const int lrow = sl >> 2;                           // a 1-KiB piece = 16 rows x 64 B: lane -> row lane >> 2, chunk slot lane & 3
    const int kc = ((sl & 3) ^ ((0x78 >> (((lrow >> 2) & 3) * 2)) & 3)) * 16;  // (bytes) the slot holds global chunk slot ^ swz(row)''')
rep('''    n0 = (c0b + rem / gmr) * 256;''','''    n0 = (c0b + rem / gmr) * 128;''')
cut('''#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int ra = min(m0 + wr * 128''','''  };
  const unsigned wdst''','''#pragma unroll
    for (int j = 0; j < 4; ++j) {                        // this wave fills A pieces 4 wid + j (rows 16 per piece) ...
      const int ra = min(m0 + (wid * 4 + j) * 16 + lrow, p.M - 1);
      oa[j] = (unsigned)((gA > 0 ? (long long)(ra / gA) * p.sgA + (long long)(ra % gA) * p.lda : (long long)ra * p.lda) * EB + kc);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {                        // ... and B pieces 2 wid + j
      const int rb = min(n0 + (wid * 2 + j) * 16 + lrow, p.N - 1);
      ob[j] = (unsigned)((long long)rb * p.ldb * EB + kc);
    }
''')
cut('''  const unsigned wdst = ''','''  // bias slice of this wave's 64 columns''','''  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  auto stage = [&](int buf, int k0) {          // one stage: 32 K elements of every row of the tile, six LDS-DMAs per wave
    const unsigned base = lds0 + buf * ST;
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16s(Ab + k0 * 2, oa[j], base + (wid * 4 + j) * 1024);
#pragma unroll
    for (int j = 0; j < 2; ++j) glds16s(Bb + k0 * 2, ob[j], base + A_ST + (wid * 2 + j) * 1024);
  };
  auto stage_k0 = [&](int) { stage(0, 0); };
''')
rep('''  const int nt = p.K / (F8 ? 2 * BK : BK);      // K-tiles of 128 B per row

  const int aoff = wr * 64 * 128, boff = wc * 32 * 128;
''','''  const int nt = p.K / BK;                      // stages of 64 B per row
''')
# stagger removal
cut('''  // Free de-phasing: when the last round''','''  setup(cur);
  stage_k0(0);''','')
# top of tile + K loop
cut('''    // Nothing but the walk position and the lane index is carried across an epilogue''','''    // ---------------- epilogue of tile (em0, en0, ez)''','''    // The first stage of this tile is in flight in ring slot 0 (issued before the loop / behind the previous epilogue); the operand
    // offsets of setup() are still live.  Waits: a wave's six DMAs of stage t are retired before the barrier that opens stage t, with
    // the six of stage t + 1 (and, at t = 0, the four bias loads behind them) left in flight.
    const bool bvec = bias_vec();
    if (nt > 1) stage(1, BK);
    fetch_bias(bvec);
    int fl = lane;
    asm volatile("" : "+v"(fl));
    const int fo = (fl & 15) * 64 + (((fl >> 4) ^ ((0x78 >> ((((fl & 15) >> 2) & 3) * 2)) & 3)) << 4);
    const int aoff = wr * 128 * 64 + fo, boff = A_ST + wc * 64 * 64 + fo;

    f32x4 acc[2][2][2][4];           // [mh][nh][ni][mi]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[a][b][c][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    int rb_ = 0;                       // ring slot of stage t
    for (int t = 0; t < nt; ++t) {
      if (t == 0) {
        if (nt > 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else if (t + 1 < nt) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      VLA_BARRIER();                   // stage t is in for every wave; every wave is done reading stage t - 1
      if (t + 2 < nt) stage(rb_ == 0 ? 2 : rb_ - 1, (t + 2) * BK);      // (t + 2) % 3 == (t - 1) % 3: the slot stage t - 1 left
      const char* kb = smem + rb_ * ST;
      bf16x8 fa[2][4], fb[2][2];
#pragma unroll
      for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[nh][ni] = *reinterpret_cast<const bf16x8*>(kb + boff + (nh * 32 + ni * 16) * 64);
#pragma unroll
      for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mh][mi] = *reinterpret_cast<const bf16x8*>(kb + aoff + (mh * 64 + mi * 16) * 64);
      VLA_MMA_QUADRANT(acc[0][0], fb[0], fa[0]);
      VLA_MMA_QUADRANT(acc[0][1], fb[1], fa[0]);
      VLA_MMA_QUADRANT(acc[1][1], fb[1], fa[1]);
      VLA_MMA_QUADRANT(acc[1][0], fb[0], fa[1]);
      rb_ = rb_ == 2 ? 0 : rb_ + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    VLA_BARRIER();                     // every operand read is finished: the ring is the epilogue's staging area now
    const int d = 0;

''')
rep('''  int d = 0;

  for (;;) {''','''  for (;;) {''')
rep('''    char* const reg = PRE ? smem + (d ^ 1) * 4 * HT + wid * STG : smem + wid * STG;''','''    char* const reg = smem + wid * STG;''')
# names
s=s.replace('gemm256_kernel','gemm_pp_kernel').replace('launch256','launch_pp')
# launcher: cut everything from 'int num_cus()' to the end and write a new one
i=s.index('int num_cus() {')
s=s[:i]+'''template <int EPI, bool RES = true, bool R2 = false>
int launch_pp(const GemmP& p0, int batch, hipStream_t st) {
  GemmP p = p0;
  p.tiles_n = (p.N + 127) / 128;
  p.ntiles = ((p.M + 255) / 256) * p.tiles_n;
  p.batch = batch;
  p.xpx = p.xpy = 0;
  p.stagger = 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_pp_kernel<EPI, false, RES, false, false, R2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_set = true;
  }
  // two workgroups per CU walk the tiles (VLA_GEMM_PP_GRID overrides the workgroup count: 0 = one workgroup per tile)
  const char* ge = getenv("VLA_GEMM_PP_GRID");
  const long long total = (long long)p.ntiles * batch;
  long long grid = ge ? atoll(ge) : 2LL * vla_num_cus();
  if (grid <= 0 || grid > total) grid = total;
  hipLaunchKernelGGL((gemm_pp_kernel<EPI, false, RES, false, false, R2>), dim3((unsigned)grid), dim3(256), LDS_BYTES, st, p);
  return 0;
}

}  // namespace

int vla_gemm_pp_launch(const GemmP& p, int epi, int batch, hipStream_t st) {
  if (p.rope_mode == 2) return launch_pp<0, false, true>(p, batch, st);
  if (epi == 1) return launch_pp<1>(p, batch, st);
  if (epi == 2) return launch_pp<2>(p, batch, st);
  return p.R ? launch_pp<0, true>(p, batch, st) : launch_pp<0, false>(p, batch, st);
}
'''
sys.stdout.write(s)
