// decoder.hip -- LatticeFasterDecoder token passing on gfx950.
//
// Replaces LatticeFasterDecoderTpl<FST, StdToken> (decoder/lattice-faster-decoder.cc):
// InitDecoding :56, AdvanceDecoding :593, GetCutoff :657, ProcessEmitting :727,
// ProcessNonemitting :833, FindOrAddToken :266, PruneForwardLinks(Final) :312/:389,
// FinalizeDecoding :638, GetRawLattice :113 -- and the HashList it runs on
// (util/hash-list-inl.h).
//
// MI355X design
//   * one decoder instance ("lane") = one persistent 1024-thread workgroup = one CU.
//     Utterances are independent, so a launch carries one workgroup per utterance and
//     the whole frame loop runs inside the kernel: no host round trip and no
//     inter-workgroup synchronisation per frame; phases are separated by workgroup
//     barriers only.  256 CUs => 256 utterances decode concurrently.
//   * HCLG in HBM as two CSRs (emitting / epsilon) of 16-byte fst::StdArc records; a
//     state's arcs are one contiguous (coalesced) run; hub states are expanded by a
//     whole wavefront (64 arcs per load instruction).
//   * state -> token map of the frame: open-addressing table of packed 64-bit
//     {state, order-preserving cost}; insert = CAS, recombination = atomicMin_u64.
//   * beam cutoff: min-reduce + counting pass; max-active / min-active by an exact
//     LDS-histogram radix select (4 x 8 bit) -- the value std::nth_element returns.
//   * token / forward-link arenas are append-only (wavefront-ballot allocation) and
//     sized for HBM3E; lattice pruning is ONE exact backward sweep at finalize (the
//     reference's periodic PruneActiveTokens is provably conservative).
//   * results are order independent and bit-exact against oracle mode 1.
#include <algorithm>
#include <vector>

#include "common.h"
#include "meta_ring.h"

namespace kamd {

typedef unsigned long long u64;
typedef unsigned int u32;

// Lane geometry (build-time): KAMD_NT threads per lane.  1024 = one lane per CU with the whole 160 KB of LDS;
// 512 = two lanes per CU (80 KB each: half the level-1 table, half the queues, a shorter LDS part of the score row):
// the search is latency-bound per lane, so two independent lanes per CU overlap each other's round trips.
#ifndef KAMD_NT
#define KAMD_NT 1024
#endif
#define NT KAMD_NT
// Minimum wavefronts per SIMD the search kernels are compiled for: 4 = a 1024-thread lane with 128 VGPRs per thread.
// tools/regime_probe.py builds with 5 (96 VGPRs) to force register spills on purpose.
#ifndef KAMD_SEARCH_WAVES_PER_EU
#define KAMD_SEARCH_WAVES_PER_EU 4
#endif
// ... and the VGPR budget the register allocator gets for them.  A 1024-thread lane may use 128; the product is built for
// KAMD_SEARCH_VGPRS = 120 and must not touch scratch memory at that (tests/test_kernel_resources.py), i.e. the kernels
// carry 8 registers of head-room against the next compiler release or source edit.  ("amdgpu_num_vgpr" counts the unified
// VGPR + AGPR file of gfx90a and later in halves: the attribute's argument is the budget divided by two.)
#ifndef KAMD_SEARCH_VGPRS
#define KAMD_SEARCH_VGPRS 120
#endif
#define KAMD_SEARCH_KERNEL __global__ __launch_bounds__(NT, KAMD_SEARCH_WAVES_PER_EU) __attribute__((amdgpu_num_vgpr(KAMD_SEARCH_VGPRS / 2)))
#define LANES_PER_CU (1024 / NT)
#define NWAVES (NT / 64)
#define EXPT 3            // tokens per thread whose records are fetched together (EXPT * NT == BIGCAP)
#define BIGCAP (EXPT * NT)   // tokens per flatten batch (deg > SMALL_DEG)
#define FIN_CAP (6 * NT)     // tokens per frame the finalize sweep keeps in LDS (6 arrays)
#define LDS_TABLE_CAP (16 * NT)  // level-1 table region, words: 128 KB of a 1024-thread lane's LDS.  A frame uses the lower half
                                 // (LDS_TABLE_SMALL words; the upper half is the commit's scratch) or, when the last frame was a
                                 // large one, all of it (the commit's lists then live in HBM): AdvanceLane
#define LDS_TABLE_SMALL (8 * NT)
#define SMALL_DEG 4
#define GL 8             // lanes of a group (4 * GL arcs of a token per trip)
#define TPG 2            // tokens a 16-lane group expands per trip (their arc records are in flight together)
#define HUGE_DEG 256      // tokens with more emitting arcs are expanded by the whole lane, one after the other
#define ARCW 4           // arcs in flight per thread in the arc-parallel expansion
#define CHUNKCAP (4 * NT)    // cached chunk owners (16 arcs each) per flatten batch
static_assert(NT == 1024 || NT == 512, "KAMD_NT: 1024 (one lane per CU) or 512 (two)");
#define EMPTY64 0xFFFFFFFFFFFFFFFFull

// threadIdx.x through a pointer-free barrier the optimiser cannot see through.  Inside a lane's frame loop everything
// that depends only on the thread id -- lane masks (tid < 256, tid < 64, ...: an SGPR pair each), tid * 12, tid & 7,
// addresses into the static LDS -- is loop invariant, and LICM hoists all of it in front of the loop: dozens of values
// live across every phase of every frame, which is what filled the register files.  Read this
// way the id is a new value at every call: what is derived from it lives where it is used.
__device__ __forceinline__ int Tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  __builtin_assume(t >= 0 && t < KAMD_NT);
  return t;
}

enum { ERR_HASH = 1, ERR_TOK = 2, ERR_LINK = 4, ERR_FRAMES = 8, ERR_WL = 16, ERR_INTERNAL = 32 };
// A state about to index the graph that is no state of the graph: the lane stops with ERR_INTERNAL instead of reading
// wherever the index points (bits 11..14 say which lookup saw it: closure, epsilon links, best token, expansion).
#define ERR_BAD_STATE(site) (ERR_INTERNAL | (256 << (site)))
#ifdef KAMD_DEBUG_OOB
#define KAMD_OOB_PRINTF(...) printf(__VA_ARGS__)
#else
#define KAMD_OOB_PRINTF(...) do {} while (0)
#endif

struct GraphDev {
  int num_states, start;
  int start_flagged;      // start | EPS_FLAG if the start state has epsilon arcs
  const uint2 *off;       // [S+1]: .x emitting arc offset, .y epsilon arc offset
  const kamd_arc *e_arcs; // emitting arcs (ilabel != 0)
  const kamd_arc *n_arcs; // epsilon arcs (ilabel == 0)
  const float *final;     // [S]
};

struct Link { int src, dst, ilabel, olabel; float graph, ac; };  // 24 B

struct LaneState {
  int frame;            // NumFramesDecoded()
  int tok_used, lnk_used;
  u32 round;
  int error;
  int finalized;
  float final_relative_cost, final_best_cost;
  int out_ntok, out_nlink;
  int out_tok_base, out_lnk_base;   // where the finalized lattice starts inside the lane's arenas
  int out_cost_in_map, presel_frames;   // staged token costs live in tok_map (FinalizeKernel2); frames with pre-selected inserts (diagnostic)
  long long counters[8];
  unsigned long long phase_cycles[16];   // diagnostic: shader cycles per phase (thread 0)
};

struct DecDev {
  GraphDev g;
  const int *tid2pdf;   // NULL => pdf = ilabel - 1
  // [emitting arcs] {weight bits, pdf}: what the cutoff test of an expanded arc needs (tid2pdf applied once).  ~90 % of the
  // expanded arcs fail that test: they cost 8 B of traffic here instead of the 16 B StdArc + 4 B pdf; the survivors'
  // records (next state, labels) are fetched by arc index in the dense insert sweep
  const uint2 *e_hot;
  int num_pdfs_lds;     // log-likelihood row entries staged in LDS per frame (0 = none)
  int lds_table_cap;    // level-1 (LDS) table words, power of two or 0
  int big_frame_tokens; // a frame after one that created more tokens than this inserts into the whole table region
  float good_first;     // > 0 (experiment, KAMD_GOOD_FIRST): tokens within this of the best are expanded in a pass of their own, first
  int preselect;        // work-queue lanes: frames with several times max_active candidates insert only those that can matter (InsertEmitted)
  int ps_margin_pct;    // pre-selection: a lane's first margin, per cent of max_active (default 8; KAMD_PS_MARGIN_PCT); adapted per frame (Sh::ps_margin_pm)
  int ps_adapt;         // the margin follows the frames (default); 0: fixed at ps_margin_pct (KAMD_PS_ADAPT=0, experiments)
  int ps_worth_pct;     // pre-selection engages when the candidates within the cutoff are at least this per cent of those under the bound (default 150; KAMD_PS_WORTH_PCT)
  int full_level2;      // every frame addresses the whole level-2 table (no per-frame mask): the second-chance launch, whose first search may have filled a frame's share
  kamd_decoder_config cfg;
  int loose;            // search mode 2: arcs are kept against the seed cutoff (kamd_decoder_set_search_mode)
  int hash_cap, hash_mask, max_frames;
  const long long *lane_tok_base, *lane_lnk_base;  // per lane: offset into the pools
  const int *lane_tok_cap, *lane_lnk_cap;          // per lane: capacity (records)
  u64 *H; u32 *slots; int *slot_tok; u32 *stamp; u32 *wl;  // per lane: hash_cap (wl: 2x)
  u64 *e2;              // per lane: hash_cap -- the frame's level-2 entries, dense, in slot-list order (CommitFrame2)
  int *tok_state; float *tok_cost; float *tok_extra; int *tok_map;  // per lane: arena_tokens
  Link *links;                                                      // per lane: arena_links
  int *tok_off;        // per lane: max_frames + 2
  int *lnk_off;        // per lane: 2 * (max_frames + 2) + 1
  float *cost_offsets; // per lane: max_frames + 1
  int *trace_ntok; float *trace_cutoff;  // per lane: max_frames + 1
  float *scratch;      // per lane: 2 * hash_cap
  LaneState *st;
};

// per-lane view
struct Ctx {
  u64 *H; u64 *e2; u32 *slots; int *slot_tok; u32 *stamp; u32 *wl0, *wl1;
  int *tok_state; float *tok_cost; float *tok_extra; int *tok_map;
  Link *links; int *tok_off; int *lnk_off; float *cost_offsets; int *trace_ntok;
  float *trace_cutoff; float *scratch; LaneState *st;
  int tok_cap, lnk_cap;
};

__device__ inline Ctx MakeCtx(const DecDev &d, int lane) {
  Ctx c;
  size_t hc = static_cast<size_t>(d.hash_cap) + LDS_TABLE_CAP, mf = d.max_frames;
  const long long tbase = d.lane_tok_base[lane], lbase = d.lane_lnk_base[lane];
  c.tok_cap = d.lane_tok_cap[lane]; c.lnk_cap = d.lane_lnk_cap[lane];
  c.H = d.H + lane * static_cast<size_t>(d.hash_cap); c.e2 = d.e2 + lane * static_cast<size_t>(d.hash_cap); c.slots = d.slots + lane * hc; c.slot_tok = d.slot_tok + lane * hc;
  c.stamp = d.stamp + lane * hc; c.wl0 = d.wl + lane * 2 * hc; c.wl1 = c.wl0 + hc;
  c.tok_state = d.tok_state + tbase; c.tok_cost = d.tok_cost + tbase;
  c.tok_extra = d.tok_extra + tbase; c.tok_map = d.tok_map + tbase;
  c.links = d.links + lbase; c.tok_off = d.tok_off + lane * (mf + 2);
  c.lnk_off = d.lnk_off + lane * (2 * (mf + 2) + 1);
  c.cost_offsets = d.cost_offsets + lane * (mf + 1);
  c.trace_ntok = d.trace_ntok + lane * (mf + 1); c.trace_cutoff = d.trace_cutoff + lane * (mf + 1);
  c.scratch = d.scratch + lane * 2 * hc; c.st = d.st + lane;
  return c;
}

// ---- phase-scoped views of the launch descriptors -------------------------------------------------------------
// A search lane keeps ~100 uniform values alive (the DecDev fields, the lane's Ctx pointers, LDS regions) and hipcc
// hoists every  base + tid * size  it can out of the loops.  Taken from the by-value kernel argument they are all live
// from the kernel's entry to its end: 105-148 SGPR spills and a VGPR file full of loop-invariant 64-bit addresses,
// i.e. a kernel one source edit away from spilling to scratch memory.  Instead every PHASE of
// a frame (cutoff, expansion, inserts, commit) and every stage of an utterance (init, finalize, hand-over) re-reads
// what it needs from the kernarg segment through a pointer the optimiser cannot see through: the s_load's (scalar
// cache hits, a few hundred cycles per phase against ~10^5 per frame) cannot be hoisted above the phase's start, so
// the descriptors -- and everything derived from them -- are live inside one phase only.
// Every kernel that uses this has the DecDev as its FIRST parameter (kernarg offset 0).
template <typename T> using KPtr = const T __attribute__((address_space(4))) *;
__device__ __forceinline__ DecDev LoadDecDev() {
  KPtr<DecDev> p = (KPtr<DecDev>)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const DecDev *)p;
}
// a second by-value parameter that directly follows the DecDev
template <typename T>
__device__ __forceinline__ T LoadSecondArg() {
  typedef const unsigned char __attribute__((address_space(4))) *KBytes;
  KBytes b = (KBytes)__builtin_amdgcn_kernarg_segment_ptr() + ((sizeof(DecDev) + alignof(T) - 1) / alignof(T)) * alignof(T);
  KPtr<T> p = (KPtr<T>)b;
  asm volatile("" : "+s"(p));
  return *(const T *)p;
}
// the same for a value the kernel already holds in SGPRs (a lane number, a task): what is derived from it after this
// point cannot be computed before it
__device__ __forceinline__ int Opaque(int v) { v = __builtin_amdgcn_readfirstlane(v); asm volatile("" : "+s"(v)); return v; }

#define SH_HIST (LDS_TABLE_CAP / 32 > NT ? LDS_TABLE_CAP / 32 : NT)
#define SH_CAND (NT / 2)
struct Sh {  // workgroup-shared state
  u64 red64[NWAVES];
  int redi[NWAVES];
  int redj[NWAVES];
  int redk[NWAVES];
  int redl[NWAVES];
  u64 best_key;          // min {ordered cost, state} of the newest token list
  int c_lt, c_le;        // #costs < / <= best+beam of the newest token list
  int n_new;
  float redf[NWAVES];
  u32 hist[SH_HIST];     // radix-select bins / one "queued" bit per level-1 slot / the commit's linear cost histogram (TblSelectLinear)
  float sel_cand[SH_CAND];   // TblSelectLinear: the members of the bucket that holds the wanted rank
  u32 next_cutoff_u;
  int n_slots, n_slots1, n_links, n_surv, n_final, wl_n[2], err, bigcnt, hugecnt, changed;   // n_links: candidates recorded; n_surv: those the insert sweep carried on
    // n_slots: level-2 (HBM) entries, n_slots1: level-1 (LDS), counted by the commit
  int cache_valid;       // the newest token list's costs are in the LDS cost cache (small-table frames)
  // the work-queue lane drops tokens that can never be expanded when it commits a list (CommitFrame2, `drop`): the list
  // then holds cur_n of the cur_n_all tokens created, and its GetCutoff has been evaluated already (on all of them)
  int cur_n_all, cutoff_ready;
  float nx_cur_cutoff, nx_adaptive_beam;
  int sel_bin, sel_below;
  int scan_total;
  int big_total;
  int presel_frames;     // frames of this call whose inserts were pre-selected (diagnostic)
  int ps_margin_pm;      // pre-selection: candidates wanted under the bound beyond max_active, per mille of max_active; adapted frame by
                         // frame to the share of candidates that turned out to be second arcs into a state (PhaseInsert)
  // per-lane running state mirrored in LDS (the global copies are written for the host and
  // for the next launch; reading them back every frame would be an L2 round trip each)
  int cur_tb, cur_n;     // newest token list: first token, count
  int lnk_used; u32 round;
  long long cnt[8];
  unsigned long long ph[16];
  unsigned long long t_prev;
};

// phase ids for the diagnostic cycle breakdown
// (the finalize sweep, per frame: FIN_FETCH = until the frame's records -- requested one frame ahead -- are in LDS,
// FIN_EMIT / FIN_EPS = the two relaxations, FIN_STAGE = survivors staged; FIN_SWEEP = what is left: HBM-mode frames, the end;
// COMMIT_SCAN = the commit's counting sweeps over the table: entries, best token, and -- `drop` -- the next frame's GetCutoff)
enum { PH_FIN_FETCH = 0, PH_CUTOFF, PH_SEED, PH_EXPAND, PH_EXPAND_BIG, PH_EPS_CLOSURE, PH_COMPACT,
       PH_FIXUP, PH_EPS_LINKS, PH_CLEAR, PH_FIN_SWEEP, PH_FIN_COMPACT, PH_COMMIT_SCAN, PH_FIN_EMIT, PH_FIN_EPS, PH_FIN_STAGE };
__device__ inline void Stamp(Sh *sh, int idx) {   // call right after a barrier
  if (Tid() == 0) {
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    sh->ph[idx] += now - sh->t_prev;
    sh->t_prev = now;
  }
}

// ---------------------------------------------------------------- primitives
__device__ inline u64 LoadH(const u64 *p) {  // L1-bypassing load: the table is written by L2 atomics
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline u32 LoadU32(const u32 *p) {  // for words updated by L2 atomics (L1 may be stale)
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// all of this wavefront's stores have reached L2 (write-through) before it continues
__device__ inline void DrainStores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Workgroup barrier that orders LDS accesses only: s_waitcnt lgkmcnt(0) + s_barrier.  A plain
// __syncthreads() is also a release of the wave's GLOBAL stores, i.e. s_waitcnt vmcnt(0) —
// which on gfx9 drains every outstanding load too (one counter): register prefetches and
// fire-and-forget stores would be waited for at every barrier.  Use only where no thread reads
// global data another thread of the workgroup wrote since the last full barrier.
__device__ inline void LdsBarrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ inline u32 HashState(int s, int mask) {
  return (static_cast<u32>(s) * 2654435761u >> 7) & static_cast<u32>(mask);
}
__device__ inline u64 Pack(int state, float cost) {
  return (static_cast<u64>(static_cast<u32>(state)) << 32) | FloatToOrdered(cost);
}
__device__ inline float CostOf(u64 e) { return OrderedToFloat(static_cast<u32>(e)); }
// table keys / device arc targets carry "this state has epsilon arcs" in bit 31
#define EPS_FLAG 0x80000000u
__device__ inline int StateOf(u64 e) { return static_cast<int>(e >> 32); }           // flagged
__device__ inline int PlainState(int flagged) { return flagged & 0x7FFFFFFF; }
__device__ inline bool HasEps(int flagged) { return (static_cast<u32>(flagged) & EPS_FLAG) != 0; }

// one slot from a workgroup counter for every ACTIVE lane of the wavefront (call under
// the predicate): wavefront-ballot aggregation => one LDS atomic per wavefront.  The leader
// is the first active lane: v_readfirstlane hands its result round (a __shfl would be a
// ds_bpermute, ~100 cycles through the LDS crossbar, on every call of the inner loops).
__device__ inline int WaveAlloc(int *counter) {
  const u64 m = __ballot(1);
  const int lane = Tid() & 63;
  const int leader = __ffsll(static_cast<long long>(m)) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

// Full-wavefront reductions on the DPP path (call with every lane active).  Two quad permutes, row_half_mirror and
// row_mirror leave the total of each row of 16 in all its lanes, four v_readlane combine the rows: ~50 cycles, against
// ~600 for the six dependent ds_bpermute of a __shfl_xor butterfly (measured on 1024-thread workgroups: a 64-bit
// workgroup min built on shuffles costs 2.1 us, four or five of those sat in every frame).
#define KAMD_DPP(x, ctrl) __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xf, 0xf, false)
#define KAMD_DPP_STEPS(OP) OP(0xB1) OP(0x4E) OP(0x141) OP(0x140)   /* quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror */
__device__ inline int WaveSumI(int v) {
#define KAMD_OP(c) v += KAMD_DPP(v, c);
  KAMD_DPP_STEPS(KAMD_OP)
#undef KAMD_OP
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ inline float WaveMinF(float v) {
#define KAMD_OP(c) v = fminf(v, __int_as_float(KAMD_DPP(__float_as_int(v), c)));
  KAMD_DPP_STEPS(KAMD_OP)
#undef KAMD_OP
  const int i = __float_as_int(v);
  return fminf(fminf(__int_as_float(__builtin_amdgcn_readlane(i, 0)), __int_as_float(__builtin_amdgcn_readlane(i, 16))),
               fminf(__int_as_float(__builtin_amdgcn_readlane(i, 32)), __int_as_float(__builtin_amdgcn_readlane(i, 48))));
}
// inclusive prefix sum over the wavefront: Hillis-Steele inside the rows of 16 (row_shr, zero fill), then the totals of
// the rows before
__device__ inline int WaveInclScanI(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
  const int row = (Tid() & 63) >> 4;
  return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}
__device__ inline u64 WaveMin64(u64 v) {
#define KAMD_OP(c) { const u64 t = (static_cast<u64>(static_cast<u32>(KAMD_DPP(static_cast<int>(v >> 32), c))) << 32) | \
                                   static_cast<u32>(KAMD_DPP(static_cast<int>(v), c)); v = t < v ? t : v; }
  KAMD_DPP_STEPS(KAMD_OP)
#undef KAMD_OP
  u64 r = ~0ull;
#pragma unroll
  for (int l = 0; l < 64; l += 16) {
    const u64 t = (static_cast<u64>(static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(v >> 32), l))) << 32) |
                  static_cast<u32>(__builtin_amdgcn_readlane(static_cast<int>(v), l));
    r = t < r ? t : r;
  }
  return r;
}

__device__ inline u64 BlockMin64(u64 v, Sh *sh) {
  v = WaveMin64(v);
  __syncthreads();
  if ((Tid() & 63) == 0) sh->red64[Tid() >> 6] = v;
  __syncthreads();
  u64 r = sh->red64[0];
  for (int i = 1; i < NWAVES; i++) r = sh->red64[i] < r ? sh->red64[i] : r;
  return r;
}
template <bool kLdsOnly = false> __device__ inline void Bar() { if (kLdsOnly) LdsBarrier(); else __syncthreads(); }
template <bool kLdsOnly = false>
__device__ inline void BlockSum2(int &a, int &b, Sh *sh) {
  a = WaveSumI(a); b = WaveSumI(b);
  Bar<kLdsOnly>();
  if ((Tid() & 63) == 0) { sh->redi[Tid() >> 6] = a; sh->redj[Tid() >> 6] = b; }
  Bar<kLdsOnly>();
  a = 0; b = 0;
  for (int i = 0; i < NWAVES; i++) { a += sh->redi[i]; b += sh->redj[i]; }
}
template <bool kLdsOnly = false>
__device__ inline void BlockSum4(int &a, int &b, int &c2, int &d2, Sh *sh) {
  a = WaveSumI(a); b = WaveSumI(b); c2 = WaveSumI(c2); d2 = WaveSumI(d2);
  Bar<kLdsOnly>();
  if ((Tid() & 63) == 0) {
    const int w = Tid() >> 6;
    sh->redi[w] = a; sh->redj[w] = b; sh->redk[w] = c2; sh->redl[w] = d2;
  }
  Bar<kLdsOnly>();
  a = 0; b = 0; c2 = 0; d2 = 0;
  for (int i = 0; i < NWAVES; i++) { a += sh->redi[i]; b += sh->redj[i]; c2 += sh->redk[i]; d2 += sh->redl[i]; }
}
template <bool kLdsOnly = false>
__device__ inline float BlockMinF(float v, Sh *sh) {
  v = WaveMinF(v);
  Bar<kLdsOnly>();
  if ((Tid() & 63) == 0) sh->redf[Tid() >> 6] = v;
  Bar<kLdsOnly>();
  float r = sh->redf[0];
  for (int i = 1; i < NWAVES; i++) r = fminf(r, sh->redf[i]);
  return r;
}
// deterministic (index ordered) exclusive scan of a 0/1 flag over the workgroup
__device__ inline int BlockScanFlag(bool f, int *total, Sh *sh) {
  const u64 m = __ballot(f);
  const int lane = Tid() & 63, w = Tid() >> 6;
  __syncthreads();
  if (lane == 0) sh->redi[w] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < NWAVES; i++) { int c = sh->redi[i]; if (i < w) base += c; tot += c; }
  *total = tot;
  return base + __popcll(m & ((1ull << lane) - 1ull));
}

// wavefront-aggregated histogram add (costs of one frame cluster in few digits)
__device__ inline void WaveHistAdd(u32 *hist, int bin, bool active) {
  u64 todo = __ballot(active);
  const int lane = Tid() & 63;
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    const int lb = __shfl(bin, leader, 64);
    const u64 same = __ballot(active && bin == lb);
    if (lane == leader) atomicAdd(&hist[lb], static_cast<u32>(__popcll(same)));
    todo &= ~same;
    if (bin == lb) active = false;
  }
}

typedef __attribute__((address_space(3))) const float lds_cfloat;

// exact k-th smallest (0-based) of cost[0..n): the value std::nth_element leaves at
// position k (lattice-faster-decoder.cc:693-697, 707-712).  4-pass 8-bit radix select
// on order-preserving keys with an LDS histogram.
template <typename SrcPtr>
__device__ __forceinline__ float BlockSelectKth(SrcPtr cost, int n, int k, Sh *sh) {
  u32 prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    __syncthreads();
    if (Tid() < 256) sh->hist[Tid()] = 0;
    __syncthreads();
    for (int base = 0; base < n; base += NT) {
      int i = base + Tid();
      bool act = false; int bin = 0;
      if (i < n) {
        u32 key = FloatToOrdered(cost[i]);
        act = (key & mask) == prefix;
        bin = (key >> shift) & 255;
      }
      WaveHistAdd(sh->hist, bin, act);
    }
    __syncthreads();
    if (Tid() < 64) {   // one wavefront: 4 bins per lane, shuffle scan, locate rank k
      const int l = Tid();
      const int h0 = sh->hist[4 * l], h1 = sh->hist[4 * l + 1], h2 = sh->hist[4 * l + 2], h3 = sh->hist[4 * l + 3];
      const int mine = h0 + h1 + h2 + h3;
      int incl = mine;
      for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (l >= o) incl += t; }
      const int excl = incl - mine;
      if (k >= excl && k < incl) {
        int cum = excl, b = 4 * l;
        if (cum + h0 > k) { } else { cum += h0; b++; if (cum + h1 > k) { } else { cum += h1; b++; if (cum + h2 > k) { } else { cum += h2; b++; } } }
        sh->sel_bin = b; sh->sel_below = cum;
      }
    }
    __syncthreads();
    prefix |= static_cast<u32>(sh->sel_bin) << shift;
    mask |= 255u << shift;
    k -= sh->sel_below;
  }
  return OrderedToFloat(prefix);
}

// max-active cutoff: the max_active-th smallest cost, known to lie below beam_cutoff.
// One pass builds a 3072-bucket LDS histogram over the linear range [best, best+beam)
// (monotone bucket map => lower buckets hold smaller costs); the bucket containing rank k
// is then resolved EXACTLY by ranking its few members against each other.  Falls back to
// the radix select when the bucket is too crowded.  `src` may point to LDS or global.
#define LHBINS (3 * NT)     // the bucket scan below takes exactly three buckets per thread (a fourth would cost the
                            // log-likelihood row 4 KB of LDS: 6000 pdfs no longer fit beside the tables)
#define LHCAND NT
template <typename SrcPtr>
__device__ __forceinline__ float BlockSelectLinear(SrcPtr src, int n, int k, float best, float beam,
                                   u32 *lh /* [LHBINS] LDS */, float *cand /* [LHCAND] LDS */, Sh *sh) {
  const float scale = static_cast<float>(LHBINS) / beam;
  for (int i = Tid(); i < LHBINS; i += NT) lh[i] = 0;
  __syncthreads();
  for (int i = Tid(); i < n; i += NT) {
    const float b = (src[i] - best) * scale;
    if (b < static_cast<float>(LHBINS)) atomicAdd(&lh[b < 0.f ? 0 : static_cast<int>(b)], 1u);
  }
  __syncthreads();
  // locate the bucket of rank k: 3 buckets per thread, workgroup scan
  {
    const int t = Tid(), lane = t & 63, w = t >> 6;
    const int h0 = lh[3 * t], h1 = lh[3 * t + 1], h2 = lh[3 * t + 2];
    const int mine = h0 + h1 + h2;
    const int incl = WaveInclScanI(mine);
    if (lane == 63) sh->redi[w] = incl;
    __syncthreads();
    int wbase = 0;
    for (int q = 0; q < w; q++) wbase += sh->redi[q];
    const int excl = wbase + incl - mine;
    if (t == 0) { sh->sel_bin = -1; sh->scan_total = 0; }
    __syncthreads();
    if (k >= excl && k < excl + mine) {
      int cum = excl, b = 3 * t, cnt = h0;
      if (cum + h0 <= k) { cum += h0; b++; cnt = h1; if (cum + h1 <= k) { cum += h1; b++; cnt = h2; } }
      sh->sel_bin = b; sh->sel_below = cum; sh->changed = cnt;
    }
    __syncthreads();
  }
  const int bin = sh->sel_bin, below = sh->sel_below, members = sh->changed;
  if (bin < 0 || members > LHCAND) return BlockSelectKth(src, n, k, sh);   // uniform decision
  // gather the bucket's members
  for (int i = Tid(); i < n; i += NT) {
    const float v = src[i];
    const float b = (v - best) * scale;
    if (b < static_cast<float>(LHBINS) && (b < 0.f ? 0 : static_cast<int>(b)) == bin) {
      const int p = WaveAlloc(&sh->scan_total);
      if (p < LHCAND) cand[p] = v;
    }
  }
  __syncthreads();
  const int m = sh->scan_total, kk = k - below;     // kk-th smallest of the m members
  __syncthreads();
  if (Tid() == 0) sh->sel_below = 0;
  __syncthreads();
  for (int i = Tid(); i < m; i += NT) {
    const float v = cand[i];
    int less = 0, eq = 0;
    for (int j = 0; j < m; j++) { const float u = cand[j]; less += u < v; eq += u == v; }
    if (kk >= less && kk < less + eq) sh->sel_below = static_cast<int>(FloatToOrdered(v));  // all writers agree
  }
  __syncthreads();
  const float ans = OrderedToFloat(static_cast<u32>(sh->sel_below));
  __syncthreads();
  return ans;
}

// Two-level state -> token table of one frame.  Level 1 lives in LDS (lcap words, probe
// window LWIN): the common case costs an LDS atomic (~100 cycles) instead of an L2 round
// trip (~1-2k cycles).  A state whose window is full goes to the HBM table (level 2).
// Entries are never removed inside a frame, so "window full / EMPTY seen" decide
// membership consistently.  Slot ids: [0, lcap) = LDS, lcap + g = global slot g.
#define LWIN 8      // words of a window scanned together; a table probes lwin = LWIN or 2 * LWIN words (the large table, which runs fuller)
struct Tbl {
  u64 *LH; int lcap; int lwin;
  int hmask;            // this frame's level-2 (HBM) table size - 1, a power of two <= hash_cap
  // the insert sweep queues every level-2 token it CREATES with the epsilon flag straight onto the closure's first worklist
  // (q_on): the closure then starts without a sweep through the level-2 slot list (a dependent pair of loads per entry)
  bool q_on; u32 *q_lds; int q_cap;
};
__device__ inline u32 HashL(int s, int lcap) { return (static_cast<u32>(s) * 2654435761u >> 9) & static_cast<u32>(lcap - 1); }
__device__ inline u64 TblLoad(const Ctx &c, const Tbl &t, int slot) {
  return slot < t.lcap ? t.LH[slot] : LoadH(&c.H[slot - t.lcap]);
}

// FindOrAddToken (lattice-faster-decoder.cc:266-306) on the frame's table.
// returns slot (or -1 on overflow); *improved = created, or strictly lowered the cost.
__device__ inline int HashInsert(const DecDev &d, const Ctx &c, Sh *sh, int state, float cost,
                                 bool *improved, const Tbl &t, bool *created = nullptr) {
  const int slot_bias = t.lcap, hmask = t.hmask;
  const u64 mine = Pack(state, cost);
  u32 h = HashState(state, hmask);
  for (int probe = 0; probe <= hmask; probe++) {
    // optimistic claim: one L2 round trip for a new token, and the returned word tells
    // whether an existing token of this state already has a cost <= ours.
    const u64 old = atomicCAS(&c.H[h], EMPTY64, mine);
    if (old == EMPTY64) {
      int idx = WaveAlloc(&sh->n_slots);
      if (idx < d.hash_cap) c.slots[idx] = h + static_cast<u32>(slot_bias); else atomicOr(&sh->err, ERR_HASH);
      if (t.q_on && HasEps(state)) {
        const int p = WaveAlloc(&sh->wl_n[0]);
        if (p < t.q_cap) t.q_lds[p] = h + static_cast<u32>(slot_bias);
        else if (p < d.hash_cap) c.wl0[p] = h + static_cast<u32>(slot_bias);
        else atomicOr(&sh->err, ERR_WL);
      }
      *improved = true;
      if (created) *created = true;
      return static_cast<int>(h);
    }
    if (StateOf(old) == state) {
      if (old <= mine) { *improved = false; return static_cast<int>(h); }   // :288 strict '>'
      const u64 prev = atomicMin(&c.H[h], mine);
      *improved = prev > mine;
      return static_cast<int>(h);
    }
    h = (h + 1) & static_cast<u32>(hmask);
    // a table that has overflowed is nearly full: every further insert would scan it end to end
    if ((probe & 63) == 63 && sh->err) break;
  }
  atomicOr(&sh->err, ERR_HASH);
  *improved = false;
  return -1;
}
// the frame's row: LDS copy when it fits (every expanded arc reads it), else global
struct LlRow { const float *g; lds_cfloat *l; int n_lds; };
__device__ inline float LogLikePdf(const LlRow &r, int pdf) {
  if (pdf < r.n_lds) return r.l[pdf];   // ds_read (a generic pointer here would be a FLAT load: vmcnt(0)+lgkmcnt(0))
  return r.g[pdf];
}

// the frame's candidate records: 16-byte words in the part of the link arena that starts at link `link_base` (n candidates
// never reach beyond link_base + n: 16 n + 8 <= 24 n)
__device__ __forceinline__ uint4 *CandBase(const Ctx &c, int link_base) {
  return reinterpret_cast<uint4 *>((reinterpret_cast<size_t>(c.links + link_base) + 15) & ~static_cast<size_t>(15));
}
__device__ inline int TblInsert(const DecDev &d, const Ctx &c, Sh *sh, const Tbl &t, int state, float cost,
                                bool *improved, bool *created = nullptr);
// W emitting arcs of one thread (lattice-faster-decoder.cc:791-809), evaluated together:
// all cutoff tests first (the arcs were loaded together), then the inserts and links.
template <int W>
__device__ inline void ProcessArcs(const DecDev &d, const Ctx &c, Sh *sh, const Tbl &tbl, const LlRow &ll,
                                   const uint2 (&hot)[W], const u32 (&aidx)[W], const int (&src_tok)[W],
                                   const float (&cur_cost)[W], const bool (&ok)[W], float cost_offset,
                                   float adaptive_beam, int link_base, bool loose, float seed_cutoff) {
  float ac[W], tot[W], llv[W];
  bool pass[W];
  // the running bound, read once for the W arcs: any value >= the frame's final cutoff is a valid filter here (the dense
  // sweep applies the final one), and the bound only ever tightens
  float nc = OrderedToFloat(sh->next_cutoff_u);
  // the W scores: LDS reads issued together, no branch in between (hipcc drains vmcnt / lgkmcnt at the end of every
  // conditional block that loads, which would make the W arcs W sequential round trips); the pdfs beyond the LDS part
  // of the row -- none at all for the usual chain model -- are patched from HBM afterwards
#pragma unroll
  for (int q = 0; q < W; q++) {
    const int pdf = ok[q] ? static_cast<int>(hot[q].y & ~EPS_FLAG) : 0;
    llv[q] = ll.l[max(min(pdf, ll.n_lds - 1), 0)];
  }
#pragma unroll
  for (int q = 0; q < W; q++)
    if (ok[q] && static_cast<int>(hot[q].y & ~EPS_FLAG) >= ll.n_lds) llv[q] = ll.g[hot[q].y & ~EPS_FLAG];
#pragma unroll
  for (int q = 0; q < W; q++) {
    ac[q] = cost_offset - llv[q];
    tot[q] = cur_cost[q] + ac[q] + __uint_as_float(hot[q].x);
    pass[q] = ok[q] && !(tot[q] > (loose ? seed_cutoff : nc));
    if (pass[q]) {
      const float cand = tot[q] + adaptive_beam;
      if (cand < nc) { atomicMin(&sh->next_cutoff_u, FloatToOrdered(cand)); nc = cand; }
    }
  }
  // candidates are only RECORDED here, as (source token, ARC INDEX, costs).  The arc's record (next state, labels) and
  // the table inserts come later, in a dense sweep over the recorded links (InsertEmitted): with ~10 % of the arcs
  // passing, fetching and inserting in place would run the expensive path at ~10 % lane utilisation.
#pragma unroll
  for (int q = 0; q < W; q++) {
    if (!pass[q]) continue;
    const int k = WaveAlloc(&sh->n_links);
    if (link_base + k >= c.lnk_cap) { atomicOr(&sh->err, ERR_LINK); continue; }
    // a candidate is 16 bytes: source token, arc index, the arc's total cost (the insert sweep then needs no look at the
    // source token) and its acoustic part; labels, target and graph weight come from the arc record, for the survivors only
    // (bit 31 of the arc word: the target state has epsilon arcs -- InsertEmitted's pre-selection)
    CandBase(c, link_base)[k] = make_uint4(static_cast<u32>(src_tok[q]), aidx[q] | (hot[q].y & EPS_FLAG), __float_as_uint(tot[q]), __float_as_uint(ac[q]));
  }
}

#define L2B 4             // level-2 (HBM) table entries / links per thread fetched together in the commit
#define INSB 4            // links per thread fetched together in InsertEmitted
static_assert(EXPT * NT == BIGCAP, "one outer expansion iteration must fit the flatten queue");
// Dense sweep over the frame's recorded emitting links (lattice-faster-decoder.cc:803-809):
// FindOrAddToken for every link whose own tot passes the FINAL next_cutoff; the link's dst
// becomes the table slot, or -1 when the arc is outside the final cutoff (the canonical
// rule: no order-dependent extras ever enter the table).
// Where a frame's links live while it is being built (round 4: every link record is written ONCE in its final form, and
// only when it is kept): the expansion records CANDIDATES {src, arc index, tot, graph, ac} at [link_begin, + n_cand); the
// (16 bytes each: CandBase); the insert sweep below reads them once and appends the SURVIVORS -- final form but for dst, which is the table slot --
// behind them, at [link_begin + n_cand, + n_surv); the commit reads those, resolves slot -> token and writes the FINAL
// links, dense from link_begin again (over the dead candidates: a final link's index is never beyond its survivor's,
// which lies behind every candidate), leaving out the ones whose destination token got no record (CommitFrame2, `drop`).
// The arena needs room for n_cand + n_surv records beyond the links in use; the records behind the final links are scratch.
// Pre-selection (round 5; work-queue lanes, frames whose candidates are several times max_active -- the frames behind a
// word boundary, 10^5 candidates of which the next frame's max-active cutoff keeps 7000).  Every token such a frame creates
// beyond the NEXT frame's cutoff is deleted unexpanded, and was until now inserted all the same: 84 % of the planted load's
// tokens, most of them in the HBM level of the table (a CAS, a slot-list word, the dense copy, the slot -> token word, the
// clearing store: five random sectors each).  Instead: a histogram of the candidates' costs picks a bound B that about
// 1.25 max_active candidates lie under; the sweep inserts the candidates with tot <= B and those whose target has epsilon
// arcs and counts the tokens it CREATES under B.  If that count exceeds max_active, the next frame's
// cutoff -- the max_active-th smallest token cost -- is <= B whatever the other candidates are, every token it can expand
// is in the table with its final cost, and the others would only have been counted: they are not inserted at all, and
// after the epsilon closure FindSkipped turns the left-out candidates whose target IS in the table into links (a link
// into a live token from a worse arc is a lattice arc like any other).  If the count falls short (many candidates of few
// states) the bound is raised and the next slab of candidates inserted, twice at most; then everything is (PhaseInsert).  Lattices, links, cutoffs and
// every work counter but one are unchanged: N_tok (counters[5], trace_ntok) counts the tokens the lane inserted.
// A sweep inserts one SLAB of the candidates: those with slab_lo < tot <= slab_hi; the candidates whose target has
// epsilon arcs belong to the first slab (slab_lo = -inf) whatever they cost.  (-inf, +inf] is every candidate (a frame
// without pre-selection), (B, +inf] what a pre-selection that proved nothing left out.  *created_in_slab: tokens this
// sweep created with a cost <= slab_hi.
__device__ inline int InsertEmitted(const DecDev &d, const Ctx &c, Sh *sh, const Tbl &tbl, int link_begin,
                                    int n_links, float cutoff, float slab_lo, float slab_hi, int *created_in_slab) {
  int k_surv = 0, made = 0;
  const int le = min(link_begin + n_links, c.lnk_cap);
  const int surv_begin = le;
  const bool first_slab = slab_lo == -INFINITY;
  // INSB links per thread per trip: the records, then the arcs, are loaded for the
  // whole batch before the first insert (two dependent round trips per batch, not per link)
  const uint4 *cand = CandBase(c, link_begin);
  const int n_cand = le - link_begin;
  for (int g0 = 0; g0 * NT < n_cand; g0 += INSB) {
    uint4 L[INSB]; kamd_arc arc[INSB]; bool take[INSB];     // L: {source token, arc index | target's epsilon flag, tot, acoustic cost}
#pragma unroll
    for (int k = 0; k < INSB; k++) {
      const int ci = Tid() + (g0 + k) * NT;
      L[k] = cand[min(ci, n_cand - 1)];          // (unconditional, clamped: see the expansion; n_cand > 0 here)
    }
#pragma unroll
    for (int k = 0; k < INSB; k++) {
      const int ci = Tid() + (g0 + k) * NT;
      const float tot = __uint_as_float(L[k].z);
      const bool eps = (L[k].y & EPS_FLAG) != 0;
      // :798 with the frame's final cutoff: a candidate beyond it is simply not carried on
      take[k] = ci < n_cand && tot <= cutoff && (first_slab ? (tot <= slab_hi || eps) : (tot > slab_lo && tot <= slab_hi && !eps));
      // the record of the arc ProcessArcs kept by index -- only for the candidates this sweep inserts (2.5 were recorded per
      // survivor at the matched load: the others all read arc 0, one cached line instead of a random 16-byte fetch each)
      arc[k] = d.g.e_arcs[take[k] ? (L[k].y & ~EPS_FLAG) : 0u];
    }
#pragma unroll
    for (int k = 0; k < INSB; k++) {
      if (!take[k]) continue;
      const float tot = __uint_as_float(L[k].z);      // == source cost + ac + graph, as ProcessArcs summed it
      bool improved, created = false;
      const int dst = TblInsert(d, c, sh, tbl, arc[k].nextstate, tot, &improved, &created);
      if (dst < 0) continue;                     // (table overflow: flagged)
      if (created && tot <= slab_hi) made++;
      k_surv++;
      const int so = surv_begin + WaveAlloc(&sh->n_surv);
      if (so >= c.lnk_cap) { atomicOr(&sh->err, ERR_LINK); continue; }
      Link o; o.src = static_cast<int>(L[k].x); o.dst = dst; o.ilabel = arc[k].ilabel; o.olabel = arc[k].olabel;
      o.graph = arc[k].weight; o.ac = __uint_as_float(L[k].w);
      c.links[so] = o;
    }
  }
  *created_in_slab = made;
  return k_surv;
}

__device__ inline int HashFind(const DecDev &d, const Ctx &c, int state, int hmask) {
  u32 h = HashState(state, hmask);
  for (int probe = 0; probe <= hmask; probe++) {
    u64 cur = LoadH(&c.H[h]);
    if (cur == EMPTY64) return -1;
    if (StateOf(cur) == state) return static_cast<int>(h);
    h = (h + 1) & static_cast<u32>(hmask);
    if ((probe & 1023) == 1023) return -1;   // only an overflowed table has runs this long
  }
  return -1;
}

// (*created, when asked for: set to true iff this call made the state's entry; left alone otherwise)
__device__ inline int TblInsert(const DecDev &d, const Ctx &c, Sh *sh, const Tbl &t, int state, float cost,
                                bool *improved, bool *created) {
  if (t.lcap > 0) {
    const u64 mine = Pack(state, cost);
    const u32 h0 = HashL(state, t.lcap), m = static_cast<u32>(t.lcap - 1);
    // scan the window LWIN words at a time with plain LDS reads (pipelined), then one atomic on the chosen word
    for (int w0 = 0; w0 < t.lwin; w0 += LWIN) {
      int w_empty = -1, w_match = -1;
#pragma unroll
      for (int w = 0; w < LWIN; w++) {
        const u64 e = t.LH[(h0 + w0 + w) & m];
        if (w_match < 0 && w_empty < 0) {
          if (e == EMPTY64) w_empty = w;
          else if (StateOf(e) == state) w_match = w;
        }
      }
      int w = w_match >= 0 ? w_match : w_empty;
      while (w >= 0 && w < LWIN) {
        const u32 sl = (h0 + w0 + w) & m;
        const u64 old = atomicCAS(&t.LH[sl], EMPTY64, mine);
        if (old == EMPTY64) { *improved = true; if (created) *created = true; return static_cast<int>(sl); }
        if (StateOf(old) == state) {
          if (old <= mine) { *improved = false; return static_cast<int>(sl); }
          const u64 prev = atomicMin(&t.LH[sl], mine);
          *improved = prev > mine;
          return static_cast<int>(sl);
        }
        w++;   // lost the word to another state: keep probing
      }
      // (these LWIN words hold other states: the next LWIN, or level 2 once the whole window is full)
    }
  }
  const int g = HashInsert(d, c, sh, state, cost, improved, t, created);
  return g < 0 ? g : g + t.lcap;
}
__device__ inline int TblFind(const DecDev &d, const Ctx &c, const Tbl &t, int state) {
  if (t.lcap > 0) {
    const u32 h0 = HashL(state, t.lcap);
    for (int w = 0; w < t.lwin; w++) {
      const u32 sl = (h0 + w) & static_cast<u32>(t.lcap - 1);
      const u64 cur = t.LH[sl];
      if (cur == EMPTY64) return -1;
      if (StateOf(cur) == state) return static_cast<int>(sl);
    }
  }
  const int g = HashFind(d, c, state, t.hmask);
  return g < 0 ? g : g + t.lcap;
}

// The candidates a pre-selected frame's insert sweeps left out (tot beyond the proven bound, no epsilon flag), after the epsilon closure: one whose
// target state is in the table all the same -- created by a better arc or by the closure -- is a link into that token
// (its cost cannot lower the token's: it is beyond the bound every inserted candidate of a non-epsilon state lies under,
// or the token is beyond the next frame's cutoff either way and gets no record); the others would have created tokens
// that nothing ever expands.  All of them passed the frame's cutoff: they count as survivors (K_surv) like before.
__device__ inline int FindSkipped(const DecDev &d, const Ctx &c, Sh *sh, const Tbl &tbl, int link_begin, int n_links, float cutoff,
                                  float ps_bound) {
  int k_surv = 0;
  const int le = min(link_begin + n_links, c.lnk_cap);
  const int surv_begin = le;
  const uint4 *cand = CandBase(c, link_begin);
  const int n_cand = le - link_begin;
  for (int g0 = 0; g0 * NT < n_cand; g0 += INSB) {
    uint4 L[INSB]; kamd_arc arc[INSB]; bool take[INSB];
#pragma unroll
    for (int k = 0; k < INSB; k++) L[k] = cand[min(Tid() + (g0 + k) * NT, n_cand - 1)];
#pragma unroll
    for (int k = 0; k < INSB; k++) {
      const float tot = __uint_as_float(L[k].z);
      take[k] = Tid() + (g0 + k) * NT < n_cand && tot <= cutoff && !(tot <= ps_bound || (L[k].y & EPS_FLAG) != 0);
      arc[k] = d.g.e_arcs[take[k] ? (L[k].y & ~EPS_FLAG) : 0u];
    }
#pragma unroll
    for (int k = 0; k < INSB; k++) {
      if (!take[k]) continue;
      k_surv++;
      const int dst = TblFind(d, c, tbl, arc[k].nextstate);
      if (dst < 0) continue;
      const int so = surv_begin + WaveAlloc(&sh->n_surv);
      if (so >= c.lnk_cap) { atomicOr(&sh->err, ERR_LINK); continue; }
      Link o; o.src = static_cast<int>(L[k].x); o.dst = dst; o.ilabel = arc[k].ilabel; o.olabel = arc[k].olabel;
      o.graph = arc[k].weight; o.ac = __uint_as_float(L[k].w);
      c.links[so] = o;
    }
  }
  return k_surv;
}

__device__ inline float LogLike(const DecDev &d, const float *ll, int ilabel) {
  // DecodableMatrixMapped::LogLikelihood (decoder/decodable-matrix.cc:62-69)
  const int pdf = d.tid2pdf ? d.tid2pdf[ilabel] : ilabel - 1;
  return ll[pdf];
}
// ProcessNonemitting (lattice-faster-decoder.cc:833-899) as a fixpoint relaxation, then
// commit the frame: compact surviving tokens into the arena, resolve emitting links,
// emit epsilon links, clear the table.  'list' is the token-list index being created.
__device__ __forceinline__ void CommitFrame(const DecDev &d, const Ctx &c, Sh *sh, const Tbl &tbl, float cutoff, int list,
                            int emit_link_begin, float *cost_cache, int cache_cap, int k_surv) {
  const int tid = Tid();
  LaneState *S = c.st;
  __syncthreads();
  // ---- epsilon closure: initial worklist = every token with epsilon arcs (:855-859)
  {
    const int ns = min(sh->n_slots, d.hash_cap);
    for (int i = tid; i < ns; i += NT) {
      const u32 slot = c.slots[i];
      const u64 e = TblLoad(c, tbl, static_cast<int>(slot));
      if (e == EMPTY64) { atomicOr(&sh->err, ERR_INTERNAL); continue; }
      if (HasEps(StateOf(e)) && CostOf(e) <= cutoff) {
        int p = WaveAlloc(&sh->wl_n[0]);
        c.wl0[p] = slot;
      }
    }
  }
  __syncthreads();
  int cur = 0;
  u32 round = sh->round;
  while (sh->wl_n[cur] > 0) {   // uniform
    round++;
    const int nw = sh->wl_n[cur];
    const u32 *wl_cur = cur ? c.wl1 : c.wl0;
    u32 *wl_nxt = cur ? c.wl0 : c.wl1;
    for (int i = tid; i < nw; i += NT) {
      const u32 slot = wl_cur[i];
      const u64 e = TblLoad(c, tbl, static_cast<int>(slot));
      const float cur_cost = CostOf(e);
      if (e == EMPTY64 || cur_cost > cutoff) continue;      // :867
      const int s = PlainState(StateOf(e));
      if (static_cast<u32>(s) >= static_cast<u32>(d.g.num_states)) { atomicOr(&sh->err, ERR_BAD_STATE(1)); continue; }
      const u32 a0 = d.g.off[s].y, a1 = d.g.off[s + 1].y;
      for (u32 a = a0; a < a1; a++) {
        const kamd_arc arc = d.g.n_arcs[a];
        const float tot_cost = cur_cost + arc.weight;
        if (tot_cost < cutoff) {            // :882
          bool improved;
          const int slot2 = TblInsert(d, c, sh, tbl, arc.nextstate, tot_cost, &improved);
          if (slot2 >= 0 && improved) {
            if (HasEps(arc.nextstate) && atomicExch(&c.stamp[slot2], round) != round) {
              int p = WaveAlloc(&sh->wl_n[cur ^ 1]);
              if (p < d.hash_cap) wl_nxt[p] = static_cast<u32>(slot2); else atomicOr(&sh->err, ERR_WL);
            }
          }
        }
      }
    }
    __syncthreads();
    const int err_now = sh->err;   // read between two barriers: uniform
    if (tid == 0) sh->wl_n[cur] = 0;
    cur ^= 1;
    LdsBarrier();
    if (err_now) break;
  }
  if (tid == 0) { sh->wl_n[0] = 0; sh->wl_n[1] = 0; }
  LdsBarrier();
  Stamp(sh, PH_EPS_CLOSURE);
  // ---- compaction: tokens with final cost <= cutoff become list 'list' (arena order is
  // the allocation order; lattices are canonicalised by (frame, state) on the host).
  // The same sweep finds the list's best token for the NEXT frame's GetCutoff.
  const int tok_base = sh->cur_tb + sh->cur_n;   // == c.tok_off[list]
  const int ns = min(sh->n_slots, d.hash_cap);
  u64 kmin = EMPTY64;
  for (int i = tid; i < ns; i += NT) {
    const u32 slot = c.slots[i];
    const u64 e = TblLoad(c, tbl, static_cast<int>(slot));
    if (e == EMPTY64) atomicOr(&sh->err, ERR_INTERNAL);   // a listed slot must hold a token
    int idx = -1;
    if (e != EMPTY64 && CostOf(e) <= cutoff) {
      idx = tok_base + WaveAlloc(&sh->n_new);
      if (idx < c.tok_cap) {
        const int st = PlainState(StateOf(e));
        c.tok_state[idx] = st;
        c.tok_cost[idx] = CostOf(e);
        c.tok_extra[idx] = 0.0f;
        if (idx - tok_base < cache_cap) cost_cache[idx - tok_base] = CostOf(e);
        if (HasEps(StateOf(e))) {           // dense list of the tokens that own epsilon arcs
          const int p = WaveAlloc(&sh->wl_n[1]);
          if (p < d.hash_cap) c.wl1[p] = static_cast<u32>(idx); else atomicOr(&sh->err, ERR_WL);
        }
        const u64 k = (static_cast<u64>(static_cast<u32>(e)) << 32) | static_cast<u32>(st);
        kmin = k < kmin ? k : kmin;
      } else { atomicOr(&sh->err, ERR_TOK); idx = -1; }
    }
    c.slot_tok[slot] = idx;
  }
  kmin = BlockMin64(kmin, sh);
  const int n_new = min(sh->n_new, c.tok_cap - tok_base);
  const float next_beam_cutoff = (n_new > 0 ? OrderedToFloat(static_cast<u32>(kmin >> 32)) : INFINITY) + d.cfg.beam;
  Stamp(sh, PH_COMPACT);
  // ---- emitting links: keep iff the arc's own tot <= final cutoff; slot -> token
  {
    const int lb = emit_link_begin, le = emit_link_begin + sh->n_links;
    for (int li = lb + tid; li < min(le, c.lnk_cap); li += NT) {
      const int slot = c.links[li].dst;
      if (slot >= 0) c.links[li].dst = c.slot_tok[slot];
    }
  }
  // ---- epsilon links of the surviving tokens (final costs), :875-897: dense over the
  // tokens that own epsilon arcs (collected during compaction)
  const int eps_link_begin = emit_link_begin + min(sh->n_links, c.lnk_cap - emit_link_begin);   // == c.lnk_off[2 * list + 1]
  int a_eps = 0, c_lt = 0, c_le = 0;
  {
    const int ne = min(sh->wl_n[1], d.hash_cap);
    for (int i = tid; i < ne; i += NT) {
      const int t = static_cast<int>(c.wl1[i]);
      const int s = c.tok_state[t];
      const float cur_cost = c.tok_cost[t];
      if (static_cast<u32>(s) >= static_cast<u32>(d.g.num_states)) { atomicOr(&sh->err, ERR_BAD_STATE(2)); continue; }
      const u32 a0 = d.g.off[s].y, a1 = d.g.off[s + 1].y;
      a_eps += static_cast<int>(a1 - a0);
      for (u32 a = a0; a < a1; a++) {
        const kamd_arc arc = d.g.n_arcs[a];
        const float tot_cost = cur_cost + arc.weight;
        if (tot_cost < cutoff) {
          const int slot2 = TblFind(d, c, tbl, arc.nextstate);
          const int dst = slot2 >= 0 ? c.slot_tok[slot2] : -1;
          if (dst < 0) { atomicOr(&sh->err, ERR_INTERNAL); continue; }
          const int li = eps_link_begin + WaveAlloc(&sh->wl_n[0]);   // worklist 0 is idle here
          if (li >= c.lnk_cap) { atomicOr(&sh->err, ERR_LINK); continue; }
          Link L; L.src = t; L.dst = dst; L.ilabel = 0; L.olabel = arc.olabel;
          L.graph = arc.weight; L.ac = 0.0f;
          c.links[li] = L;
        }
      }
    }
  }
  // ---- the next frame's GetCutoff counts, from the LDS copy of the costs
  for (int i = tid; i < n_new; i += NT) {
    const float w = i < cache_cap ? cost_cache[i] : c.tok_cost[tok_base + i];
    c_lt += w < next_beam_cutoff; c_le += w <= next_beam_cutoff;
  }
  LdsBarrier();
  Stamp(sh, PH_EPS_LINKS);
  // ---- clear the table, publish offsets and counters
  for (int i = tid; i < ns; i += NT) {
    const int sl = static_cast<int>(c.slots[i]);
    if (sl < tbl.lcap) tbl.LH[sl] = EMPTY64;
    else __hip_atomic_store(&c.H[sl - tbl.lcap], EMPTY64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  DrainStores();   // the next frame's CAS / atomicMin must find the cleared words in L2
  BlockSum4<true>(k_surv, a_eps, c_lt, c_le, sh);
  if (tid == 0) {
    const int n_eps_links = min(sh->wl_n[0], c.lnk_cap - eps_link_begin);
    c.tok_off[list + 1] = tok_base + n_new;
    c.lnk_off[2 * list + 2] = eps_link_begin + n_eps_links;
    S->tok_used = tok_base + n_new;
    S->lnk_used = eps_link_begin + n_eps_links;
    S->round = round;
    sh->round = round; sh->lnk_used = eps_link_begin + n_eps_links; sh->cur_tb = tok_base; sh->cur_n = n_new;
    sh->cur_n_all = n_new; sh->cutoff_ready = 0;
    sh->cnt[1] += a_eps;                 // A_exp: epsilon arcs of surviving tokens
    sh->cnt[3] += k_surv;                // K_surv
    sh->cnt[4] += k_surv + n_eps_links;  // L_kept
    sh->cnt[5] += n_new;                 // N_tok
    sh->best_key = kmin; sh->c_lt = c_lt; sh->c_le = c_le;
    sh->n_slots = 0; sh->n_links = 0; sh->wl_n[0] = 0; sh->wl_n[1] = 0; sh->n_new = 0;
  }
  LdsBarrier();
  Stamp(sh, PH_CLEAR);
}

// The frame's level-2 entries as the commit sees them: copied once, after the epsilon closure, from where the slot list
// points (a dependent pair of loads per entry) into a dense per-lane array.  A frame behind a word boundary can hold 10^5
// of them and the commit sweeps them several times: every later sweep is a coalesced read, E2B entries per thread in
// flight, of words this very thread wrote (entry i belongs to thread i mod NT in every sweep: no barrier in between).
#define E2B 4
template <typename F>
__device__ __forceinline__ void ForLevel2(const u64 *e2, int n, F f) {
  for (int i0 = Tid(); i0 < n; i0 += E2B * NT) {
    u64 e[E2B];
#pragma unroll
    for (int k = 0; k < E2B; k++) e[k] = e2[min(i0 + k * NT, n - 1)];
#pragma unroll
    for (int k = 0; k < E2B; k++) if (i0 + k * NT < n) f(i0 + k * NT, e[k]);
  }
}

// GetCutoff (lattice-faster-decoder.cc:657-724) on a token list of n costs, given its best cost and the two counts
// c_lt / c_le = #costs < / <= best + beam.  sel_max() / sel_min() return the max_active-th / min_active-th smallest cost
// (0-based rank: what std::nth_element leaves there) and are only called when that value decides -- uniformly.
template <typename SelMax, typename SelMin>
__device__ __forceinline__ void GetCutoff(const kamd_decoder_config &cfg, int n, float best, int c_lt, int c_le, SelMax sel_max, SelMin sel_min,
                                          float *cur_cutoff, float *adaptive_beam) {
  const float beam_cutoff = best + cfg.beam;
  if (cfg.max_active == 2147483647 && cfg.min_active == 0) { *cur_cutoff = beam_cutoff; *adaptive_beam = cfg.beam; return; }
  // nth_element(max_active) < beam_cutoff  <=>  more than max_active costs < beam_cutoff
  if (n > cfg.max_active && c_lt > cfg.max_active) {
    const float mac = sel_max();
    *adaptive_beam = mac - best + cfg.beam_delta;   // :700-702
    *cur_cutoff = mac;
    return;
  }
  float mic = INFINITY;
  if (n > cfg.min_active) {
    if (cfg.min_active == 0) mic = best;
    else if (c_le <= cfg.min_active) mic = sel_min();
    else mic = -INFINITY;  // nth_element(min_active) <= beam_cutoff: not looser than the beam
  }
  if (mic > beam_cutoff) { *adaptive_beam = mic - best + cfg.beam_delta; *cur_cutoff = mic; }  // :715-718
  else { *adaptive_beam = cfg.beam; *cur_cutoff = beam_cutoff; }
}

// BlockSelectKth over the costs of a frame's table entries where they lie: level 1 swept in LDS, the n2 level-2 entries
// through their slot list in HBM.  The same 4-pass radix select on the order-preserving cost words; no scratch but Sh.
__device__ inline float TblSelectKth(const Ctx &c, const Tbl &tbl, int n2, int k, Sh *sh) {
  const int tid = Tid();
  u32 prefix = 0, mask = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    __syncthreads();
    if (tid < 256) sh->hist[tid] = 0;
    __syncthreads();
    // (the costs of a frame share their top bytes -- a few bins per wavefront, added by ballot -- and are uniformly spread in
    // the low ones, where the ballot loop would run once per lane: plain LDS atomics there)
    auto add = [&](u64 e) {
      const u32 key = static_cast<u32>(e);
      const bool act = e != EMPTY64 && (key & mask) == prefix;
      if (shift >= 16) WaveHistAdd(sh->hist, (key >> shift) & 255, act);
      else if (act) atomicAdd(&sh->hist[(key >> shift) & 255], 1u);
    };
    for (int sl = tid; sl < tbl.lcap; sl += NT) add(tbl.LH[sl]);
    ForLevel2(c.e2, n2, [&](int, u64 e) { add(e); });
    __syncthreads();
    if (tid < 64) {   // one wavefront: 4 bins per lane, shuffle scan, locate rank k
      const int l = tid;
      const int h0 = sh->hist[4 * l], h1 = sh->hist[4 * l + 1], h2 = sh->hist[4 * l + 2], h3 = sh->hist[4 * l + 3];
      const int mine = h0 + h1 + h2 + h3;
      int incl = mine;
      for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (l >= o) incl += t; }
      const int excl = incl - mine;
      if (k >= excl && k < incl) {
        int cum = excl, b = 4 * l;
        if (cum + h0 > k) { } else { cum += h0; b++; if (cum + h1 > k) { } else { cum += h1; b++; if (cum + h2 > k) { } else { cum += h2; b++; } } }
        sh->sel_bin = b; sh->sel_below = cum;
      }
    }
    __syncthreads();
    prefix |= static_cast<u32>(sh->sel_bin) << shift;
    mask |= 255u << shift;
    k -= sh->sel_below;
  }
  __syncthreads();
  return OrderedToFloat(prefix);
}

// The max_active-th smallest cost of a frame's table entries (known to lie below best + beam), the way BlockSelectLinear
// finds it in a dense array: ONE pass builds a histogram of SH_HIST buckets over the linear range [best, best + beam)
// (monotone bucket map), the bucket that holds rank k is located by a workgroup scan, ONE more pass collects that bucket's
// members (a few dozen of 10^5) and their exact ranks decide.  Falls back to the radix select when the bucket is too
// crowded.  Level 1 swept in LDS, level 2 from its dense copy (ForLevel2): a frame of 10^5 tokens is read twice, not five
// times.  `count` is called for every entry as well (the caller's c_lt / c_le ride on the first pass).
__device__ __forceinline__ int TblBucket(float v, float best, float scale) {
  const float b = (v - best) * scale;
  return b < static_cast<float>(SH_HIST) ? (b < 0.f ? 0 : static_cast<int>(b)) : -1;
}
// pass 1: the histogram (Sh::hist), count(cost) called for every entry on the way
template <typename Count>
__device__ __forceinline__ void TblLinearHist(const Ctx &c, const Tbl &tbl, int n2, float best, float beam, Sh *sh, Count count) {
  const int tid = Tid();
  const float scale = static_cast<float>(SH_HIST) / beam;
  for (int i = tid; i < SH_HIST; i += NT) sh->hist[i] = 0;
  LdsBarrier();
  auto pass1 = [&](u64 e) {
    if (e == EMPTY64) return;
    const float v = CostOf(e);
    count(v);
    const int b = TblBucket(v, best, scale);
    if (b >= 0) atomicAdd(&sh->hist[b], 1u);
  };
  for (int sl = tid; sl < tbl.lcap; sl += NT) pass1(tbl.LH[sl]);
  ForLevel2(c.e2, n2, [&](int, u64 e) { pass1(e); });
  LdsBarrier();
}
// the rest, on the histogram TblLinearHist left in Sh::hist
__device__ inline float TblSelectLinear(const Ctx &c, const Tbl &tbl, int n2, int k, float best, float beam, Sh *sh) {
  const int tid = Tid();
  const float scale = static_cast<float>(SH_HIST) / beam;
  // Sh::redi / sel_bin / scan_total are about to be rewritten: every wavefront must have finished reading what the
  // caller's last reduction left there.  (The commit calls this right behind BlockSum2, whose readers have no barrier
  // behind them: without this one a wavefront delayed by a few hundred cycles summed a scan value instead of another
  // wavefront's count, took a different GetCutoff branch and ran the rest of the frame a barrier ahead of the others --
  // clearing the table under them.  DESIGN.md section 8.4.)
  LdsBarrier();
  // locate the bucket of rank k: SH_HIST / NT buckets per thread, workgroup scan
  constexpr int PER = SH_HIST / NT;
  static_assert(SH_HIST % NT == 0, "TblSelectLinear: whole buckets per thread");
  int h[PER], mine = 0;
#pragma unroll
  for (int q = 0; q < PER; q++) { h[q] = static_cast<int>(sh->hist[PER * tid + q]); mine += h[q]; }
  const int incl = WaveInclScanI(mine);
  if ((tid & 63) == 63) sh->redi[tid >> 6] = incl;
  if (tid == 0) { sh->sel_bin = -1; sh->scan_total = 0; }
  LdsBarrier();
  int wbase = 0;
  for (int q = 0; q < (tid >> 6); q++) wbase += sh->redi[q];
  const int excl = wbase + incl - mine;
  if (k >= excl && k < excl + mine) {
    int cum = excl, b = PER * tid;
#pragma unroll
    for (int q = 0; q < PER; q++) { if (cum + h[q] > k) break; cum += h[q]; b++; }
    sh->sel_bin = b; sh->sel_below = cum; sh->changed = static_cast<int>(sh->hist[b]);
  }
  LdsBarrier();
  const int bin = sh->sel_bin, below = sh->sel_below, members = sh->changed;
  if (bin < 0 || members > SH_CAND) return TblSelectKth(c, tbl, n2, k, sh);   // uniform decision
  auto pass2 = [&](u64 e) {
    if (e == EMPTY64) return;
    const float v = CostOf(e);
    if (TblBucket(v, best, scale) == bin) { const int p = WaveAlloc(&sh->scan_total); if (p < SH_CAND) sh->sel_cand[p] = v; }
  };
  for (int sl = tid; sl < tbl.lcap; sl += NT) pass2(tbl.LH[sl]);
  ForLevel2(c.e2, n2, [&](int, u64 e) { pass2(e); });
  LdsBarrier();
  const int m = min(sh->scan_total, SH_CAND), kk = k - below;     // kk-th smallest of the m members
  if (tid == 0) sh->sel_below = 0;
  LdsBarrier();
  for (int i = tid; i < m; i += NT) {
    const float v = sh->sel_cand[i];
    int less = 0, eq = 0;
    for (int j = 0; j < m; j++) { const float u = sh->sel_cand[j]; less += u < v; eq += u == v; }
    if (kk >= less && kk < less + eq) sh->sel_below = static_cast<int>(FloatToOrdered(v));  // all writers agree
  }
  LdsBarrier();
  const float ans = OrderedToFloat(static_cast<u32>(sh->sel_below));
  LdsBarrier();
  return ans;
}

// LDS scratch of CommitFrame2: the upper half of the table region when the frame runs on the small table; nothing
// (every cap 0: the lists then live in the lane's HBM areas) when the table takes the whole region
struct CommitLds {
  u32 *wl0, *wl1; int wl_cap;    // epsilon-closure worklists (table slots)
  uint2 *owners; int owners_cap; // {slot, cost bits} of the tokens that own epsilon arcs
  float *cost_cache; int cache_cap;   // the new token list's costs, for the next frame's GetCutoff
};

// CommitFrame for a lane with a level-1 (LDS) table: the same steps, but everything that
// concerns level-1 entries stays in LDS -- the entries are found by sweeping the table (8 or 16
// words per thread: no list of used slots is kept, an insert is one LDS atomic and nothing else),
// an entry's token index is its rank in that sweep (ballots inside a wavefront, a prefix over the
// wavefronts' counts), a committed entry's cost half is overwritten with its token index (so links resolve
// slot -> token with one ds_read), worklists and the epsilon-owner list live in idle LDS and the
// per-round "already queued" test is a bit per slot.  Level-2 (HBM) entries keep the global
// lists; when a frame has none, no barrier of the commit has to wait for global memory.
__device__ __forceinline__ void CommitFrame2(const DecDev &d, const Ctx &c, Sh *sh, const Tbl &tbl, float cutoff, int list,
                                             int emit_link_begin, int k_surv,
                                             const CommitLds &L, bool loose, bool drop, bool presel = false,
                                             float insert_cutoff = 0.0f, float presel_bound = 0.0f) {
  const int tid = Tid();
  LaneState *S = c.st;
  const int lcap = tbl.lcap;
  LdsBarrier();
  const int n2 = min(sh->n_slots, d.hash_cap);      // level-2 entries created by the emitting inserts
  if (n2 > 0) __syncthreads();                      // their slot list lives in HBM
  auto wl_put = [&](int which, int p, u32 v) { if (p < L.wl_cap) (which ? L.wl1 : L.wl0)[p] = v; else if (p < d.hash_cap) (which ? c.wl1 : c.wl0)[p] = v; else atomicOr(&sh->err, ERR_WL); };
  auto wl_get = [&](int which, int p) -> u32 { return p < L.wl_cap ? (which ? L.wl1 : L.wl0)[p] : (which ? c.wl1 : c.wl0)[p]; };
  // ---- epsilon closure: initial worklist = every token with epsilon arcs (:855-859)
  for (int sl = tid; sl < lcap; sl += NT) {
    const u64 e = tbl.LH[sl];
    if (e != EMPTY64 && HasEps(StateOf(e)) && CostOf(e) <= cutoff) wl_put(0, WaveAlloc(&sh->wl_n[0]), static_cast<u32>(sl));
  }
  // (the level-2 entries that own epsilon arcs queued themselves when the insert sweep created them: HashInsert, q_on)
  for (int i = tid; i < lcap / 32; i += NT) sh->hist[i] = 0;   // "queued this round" bits of the level-1 slots
  LdsBarrier();
  if (sh->wl_n[0] > L.wl_cap) __syncthreads();
  int cur = 0;
  u32 round = sh->round;
  while (sh->wl_n[cur] > 0) {   // uniform
    round++;
    const int nw = sh->wl_n[cur];
    for (int i = tid; i < nw; i += NT) {
      const u32 slot = wl_get(cur, i);
      const u64 e = TblLoad(c, tbl, static_cast<int>(slot));
      const float cur_cost = CostOf(e);
      if (e == EMPTY64 || cur_cost > cutoff) continue;      // :867
      const int s = PlainState(StateOf(e));
      if (static_cast<u32>(s) >= static_cast<u32>(d.g.num_states)) {
        KAMD_OOB_PRINTF("KAMD-OOB site=3 lane=%d tid=%d list=%d s=%d slot=%u e=%llx cur=%d i=%d nw=%d lcap=%d wlcap=%d\n", (int)blockIdx.x, tid, list, s, slot, (unsigned long long)e, cur, i, nw, lcap, L.wl_cap);
        atomicOr(&sh->err, ERR_BAD_STATE(3)); continue;
      }
      const u32 a0 = d.g.off[s].y, a1 = d.g.off[s + 1].y;
      // (two records per trip, fetched together: a state with an LM backoff arc and one more is the common case)
      for (u32 ab = a0; ab < a1; ab += 2) {
        const kamd_arc x0 = d.g.n_arcs[ab], x1 = d.g.n_arcs[min(ab + 1, a1 - 1)];
#pragma unroll
        for (int q = 0; q < 2; q++) {
        if (ab + q >= a1) break;
        const kamd_arc arc = q ? x1 : x0;
        const float tot_cost = cur_cost + arc.weight;
        if (tot_cost < cutoff) {            // :882
          bool improved;
          const int slot2 = TblInsert(d, c, sh, tbl, arc.nextstate, tot_cost, &improved);
          if (slot2 >= 0 && improved && HasEps(arc.nextstate)) {
            bool first;
            if (slot2 < lcap) first = (atomicOr(&sh->hist[slot2 >> 5], 1u << (slot2 & 31)) & (1u << (slot2 & 31))) == 0;
            else first = atomicExch(&c.stamp[slot2], round) != round;
            if (first) wl_put(cur ^ 1, WaveAlloc(&sh->wl_n[cur ^ 1]), static_cast<u32>(slot2));
          }
        }
        }
      }
    }
    LdsBarrier();
    if (sh->wl_n[cur ^ 1] > L.wl_cap) __syncthreads();   // uniform: the overflow of the next worklist is in HBM
    const int err_now = sh->err;     // read between two barriers: uniform
    if (tid == 0) sh->wl_n[cur] = 0;
    for (int i = tid; i < lcap / 32; i += NT) sh->hist[i] = 0;
    cur ^= 1;
    LdsBarrier();
    if (err_now) break;
  }
  if (tid == 0) { sh->wl_n[0] = 0; sh->wl_n[1] = 0; }
  // A FULL barrier in every frame (vmcnt(0) + s_barrier), not only when level-2 entries put the slot list in HBM: the
  // survivors the insert sweep appended to the link arena (InsertEmitted, compacted through a wave allocator: written by one
  // thread, resolved by another below) must have left their writers.  An LDS-only barrier orders nothing in global memory
  // on this target -- hipcc waits for vmcnt(0) at a workgroup-scope release for that reason -- and by now those stores
  // are a closure old: the wait is free.
  __syncthreads();
  Stamp(sh, PH_EPS_CLOSURE);
  // ---- a pre-selected frame: the candidates the insert sweep left out become links where their target exists
  // (FindSkipped; the table is complete now -- the closure is done -- and still holds the costs)
  if (presel) {     // uniform
    k_surv += FindSkipped(d, c, sh, tbl, emit_link_begin, sh->n_links, insert_cutoff, presel_bound);
    __syncthreads();     // the survivors it appended are resolved by other threads below
    Stamp(sh, PH_FIXUP);
  }
  // ---- compaction: every table entry becomes a token of list 'list' (all of them are within
  // the cutoff: the inserts tested it; in a loose frame -- search mode 2 -- those beyond it count too: they are just not
  // epsilon-expanded, :867).  First sweep: how many entries each thread owns and the list's best token (for the NEXT
  // frame's GetCutoff); a prefix sum turns the counts into token indices; second sweep: the records.
  const int tok_base = sh->cur_tb + sh->cur_n;   // == c.tok_off[list]
  const int ns2 = min(sh->n_slots, d.hash_cap);  // level-2 entries (emitting + closure)
  u64 kmin = EMPTY64;
  auto key_of = [&](u64 e) -> u64 { return (static_cast<u64>(static_cast<u32>(e)) << 32) | static_cast<u32>(PlainState(StateOf(e))); };
  // (token indices go to the entries in sweep order, wavefront by wavefront: within one step of a wavefront's sweep the
  // occupied lanes get consecutive indices -- a ballot -- so that the records below are written coalesced; the wavefront's
  // first index is the number of entries the wavefronts before it own)
  int wave_cnt = 0;
  for (int sl = tid; sl < lcap; sl += NT) {          // lcap is a multiple of NT: every lane of a wavefront makes the same trips
    const u64 e = tbl.LH[sl];
    const bool occ = e != EMPTY64;
    wave_cnt += __popcll(__ballot(occ));
    if (occ) { const u64 k = key_of(e); kmin = k < kmin ? k : kmin; }
  }
  // (level 2: the one sweep through the slot list -- E2B dependent pairs of loads in flight -- that leaves the entries dense)
  for (int i0 = tid; i0 < ns2; i0 += E2B * NT) {
    u32 sl[E2B]; u64 e[E2B];
#pragma unroll
    for (int k = 0; k < E2B; k++) sl[k] = c.slots[min(i0 + k * NT, ns2 - 1)];
#pragma unroll
    for (int k = 0; k < E2B; k++) e[k] = LoadH(&c.H[static_cast<int>(sl[k]) - lcap]);
#pragma unroll
    for (int k = 0; k < E2B; k++)
      if (i0 + k * NT < ns2) {
        c.e2[i0 + k * NT] = e[k];
        if (e[k] != EMPTY64) { const u64 kk = key_of(e[k]); kmin = kk < kmin ? kk : kmin; }
      }
  }
  kmin = WaveMin64(kmin);
  LdsBarrier();
  if ((tid & 63) == 0) { sh->red64[tid >> 6] = kmin; sh->redi[tid >> 6] = wave_cnt; }
  LdsBarrier();
  kmin = sh->red64[0];
  int n1 = 0, my_base = 0;                                      // level-1 entries (emitting + closure); those of the wavefronts before mine
  for (int i = 0; i < NWAVES; i++) {
    if (i > 0) kmin = sh->red64[i] < kmin ? sh->red64[i] : kmin;
    const int cw = sh->redi[i];
    if (i < (tid >> 6)) my_base += cw;
    n1 += cw;
  }
  const int n_all = n1 + ns2;
  const float best_next = n_all > 0 ? OrderedToFloat(static_cast<u32>(kmin >> 32)) : INFINITY;
  const float next_beam_cutoff = best_next + d.cfg.beam;
  // ---- `drop` (the work-queue lane, when another frame follows): a token that the NEXT frame will not expand -- its cost is
  // beyond that frame's GetCutoff, which depends on nothing but the costs in this table -- and whose state has no epsilon
  // arc has no forward link and never will have: PruneForwardLinks gives it extra cost +inf and PruneTokensForFrame deletes
  // it (lattice-faster-decoder.cc:312-383, 492-511), whatever comes later.  Such tokens get no record here and the links
  // into them are dropped like the arcs outside the final cutoff: the lane writes, keeps and finalizes what can reach the
  // lattice.  They still count (trace, counters, the next frame's max-active / min-active logic: all of that is evaluated
  // right here, on the whole table).  Not on a call's last frame, whose tokens all stay (final costs / the next call).
  const bool do_drop = drop && n_all > 0;
  float nx_cutoff = INFINITY, nx_abeam = d.cfg.beam;
  int live_base = my_base, n_live1 = n1;
  auto is_live = [&](u64 e) -> bool { return CostOf(e) <= nx_cutoff || HasEps(StateOf(e)); };
  // (a frame of no more than max_active tokens, all of them inside the next frame's beam -- the insert bound `cutoff` is not
  // beyond it -- needs neither counts nor a select: that frame's cutoff is its beam cutoff and every token is live)
  const bool all_live = do_drop && !loose && n_all <= d.cfg.max_active && n_all > d.cfg.min_active && cutoff <= next_beam_cutoff;
  if (all_live) { nx_cutoff = next_beam_cutoff; nx_abeam = d.cfg.beam; }
  if (do_drop && !all_live) {
    int cl = 0, ce = 0;
    TblLinearHist(c, tbl, ns2, best_next, d.cfg.beam, sh, [&](float w) { cl += w < next_beam_cutoff; ce += w <= next_beam_cutoff; });
    BlockSum2<true>(cl, ce, sh);
    GetCutoff(d.cfg, n_all, best_next, cl, ce, [&]() { return TblSelectLinear(c, tbl, ns2, d.cfg.max_active, best_next, d.cfg.beam, sh); },
              [&]() { return TblSelectKth(c, tbl, ns2, d.cfg.min_active, sh); }, &nx_cutoff, &nx_abeam);
    int wave_live = 0;
    for (int sl = tid; sl < lcap; sl += NT) {
      const u64 e = tbl.LH[sl];
      wave_live += __popcll(__ballot(e != EMPTY64 && is_live(e)));
    }
    LdsBarrier();
    if ((tid & 63) == 0) sh->redi[tid >> 6] = wave_live;
    LdsBarrier();
    live_base = 0; n_live1 = 0;
    for (int i = 0; i < NWAVES; i++) { const int cw = sh->redi[i]; if (i < (tid >> 6)) live_base += cw; n_live1 += cw; }
  }
  Stamp(sh, PH_COMMIT_SCAN);      // (every path above ends on a barrier)
  int a_eps = 0, c_lt = 0, c_le = 0, eps_dropped = 0;
  auto commit_entry = [&](u64 e, int pos, int *idx_out) {
    int idx = -1;
    if (loose || CostOf(e) <= cutoff) {
      idx = tok_base + pos;
      if (idx < c.tok_cap) {
        const float w = CostOf(e);
        c.tok_state[idx] = PlainState(StateOf(e));
        c.tok_cost[idx] = w;
        if (!do_drop && pos < L.cache_cap) L.cost_cache[pos] = w;          // (`drop`: that GetCutoff is done)
        c_lt += w < next_beam_cutoff; c_le += w <= next_beam_cutoff;      // the next frame's GetCutoff counts
      } else { atomicOr(&sh->err, ERR_TOK); idx = -1; }
    } else atomicOr(&sh->err, ERR_INTERNAL);     // cannot happen: the index space above counts every entry
    *idx_out = idx;
  };
  auto add_owner = [&](u32 slot, u64 e) {
    const int p = WaveAlloc(&sh->wl_n[1]);
    if (p < L.owners_cap) L.owners[p] = make_uint2(slot, static_cast<u32>(e));
    else if (p < d.hash_cap) { c.wl1[p] = slot; c.scratch[p] = CostOf(e); }
    else atomicOr(&sh->err, ERR_WL);
  };
  {
    int run = live_base;
    const u64 lt = (1ull << (tid & 63)) - 1ull;
    for (int sl = tid; sl < lcap; sl += NT) {
      const u64 e = tbl.LH[sl];
      const bool occ = e != EMPTY64;
      const bool lv = occ && (!do_drop || is_live(e));
      const u64 m = __ballot(lv);
      const int pos = run + __popcll(m & lt);
      run += __popcll(m);
      if (!occ) continue;
      int idx = -1;
      if (lv) {
        commit_entry(e, pos, &idx);
        if (idx >= 0 && HasEps(StateOf(e)) && CostOf(e) <= cutoff) add_owner(static_cast<u32>(sl), e);
      }
      tbl.LH[sl] = (e & 0xFFFFFFFF00000000ull) | static_cast<u32>(idx);   // cost half -> token index (all ones: no token)
    }
  }
  for (int i0 = tid; i0 < ns2; i0 += E2B * NT) {
    u32 sl[E2B]; u64 e[E2B];
#pragma unroll
    for (int k = 0; k < E2B; k++) { sl[k] = c.slots[min(i0 + k * NT, ns2 - 1)]; e[k] = c.e2[min(i0 + k * NT, ns2 - 1)]; }
#pragma unroll
    for (int k = 0; k < E2B; k++) {
      const int i = i0 + k * NT;
      if (i < ns2) {
        if (e[k] == EMPTY64) atomicOr(&sh->err, ERR_INTERNAL);
        else {
          int idx = -1;
          if (!do_drop) commit_entry(e[k], n1 + i, &idx);
          else if (is_live(e[k])) commit_entry(e[k], n_live1 + WaveAlloc(&sh->n_new), &idx);   // (n_new: idle since InitSh / the last commit)
          if (idx >= 0 && HasEps(StateOf(e[k])) && CostOf(e[k]) <= cutoff) add_owner(sl[k], e[k]);
          c.slot_tok[sl[k]] = idx;
        }
      }
    }
  }
  // the links below read the token indices other threads have just written: LDS only, unless level-2 entries or an
  // overflowing owner list put data in HBM that others read
  LdsBarrier();
  const int n_owner = sh->wl_n[1];
  const int n_new = min(do_drop ? n_live1 + sh->n_new : n_all, c.tok_cap - tok_base);     // the records of list 'list'
  const bool hbm_lists = ns2 > 0 || n_owner > L.owners_cap;
  if (hbm_lists) __syncthreads();
  Stamp(sh, PH_COMPACT);
  auto tok_of_slot = [&](int slot) -> int {
    if (slot < lcap) return static_cast<int>(static_cast<u32>(tbl.LH[slot]));
    return c.slot_tok[slot];
  };
  // ---- emitting links: the survivors the insert sweep left behind the candidates become the frame's final links, dense from
  // emit_link_begin, slot -> token; a link into a token without a record (`drop`) is not written at all
  {
    const int n_cand = min(sh->n_links, c.lnk_cap - emit_link_begin);
    const int surv_begin = emit_link_begin + n_cand;
    const int n_surv = min(sh->n_surv, c.lnk_cap - surv_begin);
    for (int i0 = tid; i0 < n_surv; i0 += L2B * NT) {
      Link Lk[L2B]; int t1[L2B], t2[L2B];
#pragma unroll
      for (int k = 0; k < L2B; k++) Lk[k] = c.links[surv_begin + min(i0 + k * NT, n_surv - 1)];
      // both levels are read for every link (level 1: LDS; level 2: HBM, a dummy word for the links that are not there)
#pragma unroll
      for (int k = 0; k < L2B; k++) {
        const int slot = Lk[k].dst;
        t1[k] = lcap > 0 ? static_cast<int>(static_cast<u32>(tbl.LH[min(max(slot, 0), lcap - 1)])) : -1;
        t2[k] = c.slot_tok[slot >= lcap ? slot : lcap];
      }
#pragma unroll
      for (int k = 0; k < L2B; k++) {
        const int tok = Lk[k].dst < lcap ? t1[k] : t2[k];
        if (i0 + k * NT < n_surv && tok >= 0) {
          Link o; o.src = Lk[k].src; o.dst = tok; o.ilabel = Lk[k].ilabel; o.olabel = Lk[k].olabel; o.graph = Lk[k].graph; o.ac = Lk[k].ac;
          c.links[emit_link_begin + WaveAlloc(&sh->n_final)] = o;     // (never beyond the survivor's own index: behind no candidate)
        }
      }
    }
  }
  LdsBarrier();      // every survivor has been read (the epsilon links may land on them), the count is final
  // ---- epsilon links of the surviving tokens (final costs), :875-897
  const int eps_link_begin = emit_link_begin + sh->n_final;   // == c.lnk_off[2 * list + 1]
  {
    const int ne = min(n_owner, d.hash_cap);
    for (int i = tid; i < ne; i += NT) {
      u32 slot; float cur_cost;
      if (i < L.owners_cap) { const uint2 o = L.owners[i]; slot = o.x; cur_cost = OrderedToFloat(o.y); }
      else { slot = c.wl1[i]; cur_cost = c.scratch[i]; }
      const u64 e = TblLoad(c, tbl, static_cast<int>(slot));
      const int s = PlainState(StateOf(e));
      const int t = tok_of_slot(static_cast<int>(slot));
      if (static_cast<u32>(s) >= static_cast<u32>(d.g.num_states)) {
        KAMD_OOB_PRINTF("KAMD-OOB site=4 lane=%d tid=%d list=%d s=%d slot=%u e=%llx i=%d n_owner=%d owners_cap=%d lcap=%d ns2=%d drop=%d all_live=%d t=%d n_all=%d\n", (int)blockIdx.x, tid, list, s, slot, (unsigned long long)e, i, n_owner, L.owners_cap, lcap, ns2, (int)do_drop, (int)all_live, t, n_all);
        atomicOr(&sh->err, ERR_BAD_STATE(4)); continue;
      }
      const u32 a0 = d.g.off[s].y, a1 = d.g.off[s + 1].y;
      a_eps += static_cast<int>(a1 - a0);
      for (u32 ab = a0; ab < a1; ab += 2) {
        const kamd_arc x0 = d.g.n_arcs[ab], x1 = d.g.n_arcs[min(ab + 1, a1 - 1)];
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if (ab + q >= a1) break;
          const kamd_arc arc = q ? x1 : x0;
          const float tot_cost = cur_cost + arc.weight;
          if (tot_cost < cutoff) {
            const int slot2 = TblFind(d, c, tbl, arc.nextstate);
            const int dst = slot2 >= 0 ? tok_of_slot(slot2) : -1;
            if (dst < 0) {           // `drop`: a link into a token without a record is a link the final sweep would excise
              if (do_drop && slot2 >= 0) eps_dropped++; else atomicOr(&sh->err, ERR_INTERNAL);
              continue;
            }
            const int li = eps_link_begin + WaveAlloc(&sh->wl_n[0]);   // worklist 0 is idle here
            if (li >= c.lnk_cap) { atomicOr(&sh->err, ERR_LINK); continue; }
            Link Lk; Lk.src = t; Lk.dst = dst; Lk.ilabel = 0; Lk.olabel = arc.olabel;
            Lk.graph = arc.weight; Lk.ac = 0.0f;
            c.links[li] = Lk;
          }
        }
      }
    }
  }
  LdsBarrier();
  Stamp(sh, PH_EPS_LINKS);
  // ---- clear the table, publish offsets and counters
  for (int sl = tid; sl < lcap; sl += NT) tbl.LH[sl] = EMPTY64;
  if (ns2 > 0) {
    for (int i0 = tid; i0 < ns2; i0 += E2B * NT) {
      int sl[E2B];
#pragma unroll
      for (int k = 0; k < E2B; k++) sl[k] = static_cast<int>(c.slots[min(i0 + k * NT, ns2 - 1)]);
#pragma unroll
      for (int k = 0; k < E2B; k++)
        if (i0 + k * NT < ns2) __hip_atomic_store(&c.H[sl[k] - lcap], EMPTY64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    DrainStores();   // the next frame's CAS / atomicMin must find the cleared words in L2
  }
  // the two work counters need no workgroup total in this frame: one LDS atomic per wavefront onto the running sums
  {
    const int ks = WaveSumI(k_surv), ae = WaveSumI(a_eps), ed = WaveSumI(eps_dropped);
    if ((tid & 63) == 0) {
      atomicAdd(reinterpret_cast<unsigned long long *>(&sh->cnt[1]), static_cast<unsigned long long>(ae));   // A_exp: epsilon arcs of surviving tokens
      atomicAdd(reinterpret_cast<unsigned long long *>(&sh->cnt[3]), static_cast<unsigned long long>(ks));   // K_surv
      atomicAdd(reinterpret_cast<unsigned long long *>(&sh->cnt[4]), static_cast<unsigned long long>(ks + ed));   // L_kept (+ the epsilon links below; `drop`: the unwritten ones count)
    }
  }
  BlockSum2<true>(c_lt, c_le, sh);
  if (tid == 0) {
    const int n_eps_links = min(sh->wl_n[0], c.lnk_cap - eps_link_begin);
    c.tok_off[list + 1] = tok_base + n_new;
    c.lnk_off[2 * list + 1] = eps_link_begin;              // the emitting links into list 'list' end here (only the kept ones are there)
    c.lnk_off[2 * list + 2] = eps_link_begin + n_eps_links;
    S->tok_used = tok_base + n_new;
    S->lnk_used = eps_link_begin + n_eps_links;
    S->round = round;
    sh->round = round; sh->lnk_used = eps_link_begin + n_eps_links; sh->cur_tb = tok_base; sh->cur_n = n_new;
    sh->cur_n_all = do_drop ? n_all : n_new; sh->cutoff_ready = do_drop; sh->nx_cur_cutoff = nx_cutoff; sh->nx_adaptive_beam = nx_abeam;
    atomicAdd(reinterpret_cast<unsigned long long *>(&sh->cnt[4]), static_cast<unsigned long long>(n_eps_links));
    sh->cnt[5] += do_drop ? n_all : n_new;   // N_tok: tokens created
    sh->cnt[7] += ns2;                   // (diagnostic, not part of the parity contract) tokens that went to the level-2 table
    sh->best_key = kmin; sh->c_lt = c_lt; sh->c_le = c_le;
    sh->cache_valid = !do_drop && n_new <= L.cache_cap;
    sh->n_slots = 0; sh->n_slots1 = 0; sh->n_links = 0; sh->n_surv = 0; sh->n_final = 0; sh->wl_n[0] = 0; sh->wl_n[1] = 0; sh->n_new = 0;
  }
  // the next frame's GetCutoff may read the list's costs where they lie (no LDS copy): a full barrier then
  if (!do_drop && n_new > L.cache_cap) { DrainStores(); __syncthreads(); } else LdsBarrier();
  Stamp(sh, PH_CLEAR);
}

// best token and beam counts of token list 'list' (what CommitFrame leaves behind), for
// the first frame of a launch
__device__ __forceinline__ void ComputeFrameStats(const DecDev &d, const Ctx &c, Sh *sh, int list, float *cost_cache,
                                  int cache_cap) {
  const int tb = c.tok_off[list], n = c.tok_off[list + 1] - tb;
  u64 key = EMPTY64;
  for (int i = Tid(); i < n; i += NT) {
    const u64 k = (static_cast<u64>(FloatToOrdered(c.tok_cost[tb + i])) << 32) | static_cast<u32>(c.tok_state[tb + i]);
    key = k < key ? k : key;
  }
  key = BlockMin64(key, sh);
  const float bc = (n > 0 ? OrderedToFloat(static_cast<u32>(key >> 32)) : INFINITY) + d.cfg.beam;
  int c_lt = 0, c_le = 0;
  for (int i = Tid(); i < n; i += NT) {
    const float w = c.tok_cost[tb + i];
    if (i < cache_cap) cost_cache[i] = w;
    c_lt += w < bc; c_le += w <= bc;
  }
  BlockSum2(c_lt, c_le, sh);
  if (Tid() == 0) { sh->best_key = key; sh->c_lt = c_lt; sh->c_le = c_le; sh->cache_valid = n <= cache_cap; sh->cur_n_all = n; sh->cutoff_ready = 0; }
  __syncthreads();
}

// ComputeFinalCosts (lattice-faster-decoder.cc:549-590) over token list 'list'
__device__ __forceinline__ void FinalCosts(const DecDev &d, const Ctx &c, Sh *sh, int list, float *best_cost,
                           float *best_with_final) {
  const int tb = c.tok_off[list], te = c.tok_off[list + 1];
  float b = INFINITY, bf = INFINITY;
  for (int t = tb + Tid(); t < te; t += NT) {
    const float cost = c.tok_cost[t];
    const float fc = d.g.final[c.tok_state[t]];
    b = fminf(b, cost);
    bf = fminf(bf, cost + fc);
  }
  *best_cost = BlockMinF(b, sh);
  *best_with_final = BlockMinF(bf, sh);
}

__device__ __forceinline__ void PublishLaneEnd(const DecDev &d, const Ctx &c, Sh *sh, int frame) {
  float b, bf;
  FinalCosts(d, c, sh, frame, &b, &bf);
  LaneState *S = c.st;
  const int tid = Tid();
  if (tid == 0) {
    S->frame = frame;
    S->final_relative_cost = (b == INFINITY && bf == INFINITY) ? INFINITY : bf - b;  // :574-582
    S->error |= sh->err;
    S->presel_frames += sh->presel_frames;
  }
  // (one counter per thread: unrolled on thread 0 the 24 read-modify-writes were all in flight together, 50 VGPRs)
  if (tid < 8) S->counters[tid] += sh->cnt[tid];
  if (tid < 16) S->phase_cycles[tid] += sh->ph[tid];
}

__device__ inline void InitSh(Sh *sh) {
  if (Tid() == 0) {
    sh->n_slots = 0; sh->n_slots1 = 0; sh->n_links = 0; sh->n_surv = 0; sh->n_final = 0; sh->wl_n[0] = 0; sh->wl_n[1] = 0; sh->err = 0;
    sh->bigcnt = 0; sh->hugecnt = 0; sh->n_new = 0; sh->best_key = EMPTY64; sh->c_lt = 0; sh->c_le = 0; sh->next_cutoff_u = FloatToOrdered(INFINITY);
    sh->cur_tb = 0; sh->cur_n = 0; sh->lnk_used = 0; sh->round = 0; sh->cur_n_all = 0; sh->cutoff_ready = 0; sh->cache_valid = 0; sh->presel_frames = 0; sh->ps_margin_pm = -1;
    for (int i = 0; i < 8; i++) sh->cnt[i] = 0;
    for (int i = 0; i < 16; i++) sh->ph[i] = 0;
    sh->t_prev = __builtin_amdgcn_s_memtime();
  }
  __syncthreads();
}

// pdf of every emitting arc, computed once per decoder (removes the dependent
// tid -> pdf gather of TransitionIdToPdfFast from the per-arc critical path)
__global__ void ArcHotKernel(const kamd_arc *arcs, long long n, const int *tid2pdf, uint2 *e_hot) {
  long long i = static_cast<long long>(blockIdx.x) * blockDim.x + Tid();
  // (.y: the pdf, and in bit 31 the "target state has epsilon arcs" flag the arc's nextstate carries: the insert sweep's
  // pre-selection must know it without fetching the arc record)
  if (i < n) {
    const kamd_arc a = arcs[i];
    e_hot[i] = make_uint2(__float_as_uint(a.weight), static_cast<u32>(tid2pdf ? tid2pdf[a.ilabel] : a.ilabel - 1) | (static_cast<u32>(a.nextstate) & EPS_FLAG));
  }
}

// ------------------------------------------------------------------ kernels
// InitDecoding (lattice-faster-decoder.cc:56-73): start token + ProcessNonemitting(beam)
__device__ __forceinline__ void InitLane(const DecDev &d, const Ctx &c, Sh *shp) {
  Sh &sh = *shp;
  InitSh(&sh);
  LaneState *S = c.st;
  if (Tid() == 0) {
    S->frame = 0; S->tok_used = 0; S->lnk_used = 0; S->error = 0; S->finalized = 0;
    S->final_relative_cost = INFINITY; S->final_best_cost = INFINITY;
    S->out_ntok = 0; S->out_nlink = 0; S->out_tok_base = 0; S->out_lnk_base = 0; S->presel_frames = 0;
    for (int i = 0; i < 8; i++) S->counters[i] = 0;
    for (int i = 0; i < 16; i++) S->phase_cycles[i] = 0;
    c.tok_off[0] = 0; c.lnk_off[0] = 0; c.lnk_off[1] = 0;
    sh.round = S->round;   // stamps persist across utterances: never reset
    bool imp;
    Tbl t0; t0.LH = NULL; t0.lcap = 0; t0.lwin = 0; t0.hmask = d.hash_mask; t0.q_on = false; t0.q_lds = NULL; t0.q_cap = 0;
    HashInsert(d, c, &sh, d.g.start_flagged, 0.0f, &imp, t0);
  }
  __syncthreads();
  Tbl tbl; tbl.LH = NULL; tbl.lcap = 0; tbl.lwin = 0; tbl.hmask = d.hash_mask; tbl.q_on = false; tbl.q_lds = NULL; tbl.q_cap = 0;     // InitDecoding has no LDS table: level 2 only
  CommitFrame(d, c, &sh, tbl, d.cfg.beam, 0, 0, NULL, 0, 0);
  PublishLaneEnd(d, c, &sh, 0);
}
__global__ __launch_bounds__(NT, 4) void InitKernel(DecDev d, const int *lanes) {
  __shared__ Sh sh;
  const Ctx c = MakeCtx(d, lanes[blockIdx.x]);
  InitLane(d, c, &sh);
}

// The lane's regions of the dynamic LDS (a function of two DecDev fields: recomputed by every phase from its own view)
struct AdvLds {
  float *ll;         // [num_pdfs_lds] the frame's log-likelihood row (first: the DMA's LDS base must stay below 64 KB)
  u64 *T;            // [lds_table_cap] the table region; its upper half doubles as scratch:
  int2 *big_ta;      //   expansion: [BIGCAP] {token (index in list), first emitting arc}: one ds_read_b64
  int *big_scan;     //   expansion: [BIGCAP] degree
  float *cost_cache; //   GetCutoff / commit (small table): [3 * BIGCAP] the newest token list's costs (same words as the queue)
  u32 *lh;           //   GetCutoff: [LHBINS] select histogram; commit (small table): the epsilon owners
  float *cand;       //   GetCutoff: [LHCAND]
  int cap_small, cap_big;
};
// dyn_lds = [16 B][row][table region]; scratch inside the region's upper half: [A: 3 * BIGCAP words][C: LHBINS + LHCAND words]
#define ADV_SCRATCH_WORDS (3 * BIGCAP + (LHBINS + LHCAND > CHUNKCAP ? LHBINS + LHCAND : CHUNKCAP))
static_assert(ADV_SCRATCH_WORDS * 4 <= (LDS_TABLE_CAP - LDS_TABLE_SMALL) * 8, "the phases' scratch must fit the upper half of the table region");
__device__ __forceinline__ AdvLds MakeAdvLds(unsigned char *dyn_lds, int num_pdfs_lds, int lds_table_cap) {
  AdvLds a;
  a.ll = reinterpret_cast<float *>(dyn_lds + 16);
  a.T = reinterpret_cast<u64 *>(a.ll + ((num_pdfs_lds + 3) & ~3));
  a.cap_big = lds_table_cap; a.cap_small = lds_table_cap / 2;
  unsigned char *up = reinterpret_cast<unsigned char *>(a.T + a.cap_small);
  a.big_ta = reinterpret_cast<int2 *>(up);
  a.big_scan = reinterpret_cast<int *>(a.big_ta + BIGCAP);
  a.cost_cache = reinterpret_cast<float *>(up);
  a.lh = reinterpret_cast<u32 *>(up) + 3 * BIGCAP;
  a.cand = reinterpret_cast<float *>(a.lh + LHBINS);
  return a;
}
// this frame's table: all of the region after a frame that created many tokens (the level-1 table then runs at the
// load the small one has on ordinary frames, probing a window twice as long), else the lower half
#define BIG_FRAME_TOKENS (6 * NT)
__device__ __forceinline__ Tbl FrameTable(const AdvLds &a, bool big, int hmask) {
  Tbl t; t.LH = a.T; t.lcap = big ? a.cap_big : a.cap_small; t.lwin = big ? 2 * LWIN : LWIN; t.hmask = hmask;
  t.q_on = false; t.q_lds = NULL; t.q_cap = 0;
  return t;
}
// This frame's level-2 table: the lane's HBM table addressed through a mask sized for the frame -- four slots per recorded
// candidate (every token of the frame comes from one, the epsilon closure adds a few), at least L2_MIN_SLOTS, at most all of
// it.  An ordinary frame's overflow then lives in 512 KB per lane (Infinity-Cache resident for all lanes together) instead
// of being scattered over the 8 MB that the frames behind a word boundary need; the table is empty between frames, so
// every frame may pick its own size.
#define L2_MIN_SLOTS (1 << 16)
__device__ __forceinline__ int FrameLevel2Mask(int n_candidates, int hash_cap) {
  int cap = L2_MIN_SLOTS;
  const long long want = 4ll * (static_cast<long long>(n_candidates) + 4096);
  while (cap < want && cap < hash_cap) cap <<= 1;
  return min(cap, hash_cap) - 1;
}
// The frame's log-likelihood row travels HBM -> LDS by DMA (global_load_lds_dword: no registers held), issued one
// frame ahead, as soon as the expansion that reads the previous row is over: a cold 24 KB read off the critical path.
// Through inline asm, like the GEMM's ring: the compiler's own tracking of the builtin would drain vmcnt before every
// later ds_read.  The wave's LDS window goes in M0, the lane's word follows from its id.
__device__ __forceinline__ void RowDma(float *ll_lds, int num_pdfs_lds, const float *src) {
  typedef __attribute__((address_space(3))) unsigned char lds_byte;
  const unsigned ll_lds_addr = static_cast<unsigned>(reinterpret_cast<size_t>((lds_byte *)ll_lds));
  const int tid = Tid();
  for (int k0 = 0; k0 < num_pdfs_lds; k0 += NT) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane(ll_lds_addr + static_cast<unsigned>(k0 + (tid & ~63)) * 4u);
    if (k0 + tid < num_pdfs_lds)
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(m0v), "v"(src + k0 + tid) : "memory");
  }
}

// what one phase of a frame hands to the next (registers; everything else is re-read from the descriptors)
struct FrameCtl {
  int tb, n, n_all;                // the newest token list: first token, records, tokens created (>= records: CommitFrame2's `drop`)
  float best; int best_state;      // its best token
  float cur_cutoff, adaptive_beam; // GetCutoff
  float cost_offset, seed_cutoff, next_cutoff;
  bool loose;
  bool big;                        // the frame inserts into the whole table region (FrameTable)
  int hmask;                       // ... and into a level-2 table of hmask + 1 slots (FrameLevel2Mask)
  int link_base, k_surv;
  bool presel; float presel_bound; // the insert sweep left the candidates beyond presel_bound out (InsertEmitted<KAMD_PS_FIRST>): FindSkipped after the closure
};

// ---- GetCutoff (:657-724).  The best token (ties -> smallest state, as oracle mode 1) and the two beam counts were
// left behind by the sweep that created this token list (CommitFrame / ComputeFrameStats), together with an LDS copy
// of the costs.
__device__ __forceinline__ void PhaseCutoff(int lane, Sh &sh, unsigned char *dyn_lds, FrameCtl &fc) {
  const DecDev d = LoadDecDev();
  const kamd_decoder_config cfg = d.cfg;
  const AdvLds L = MakeAdvLds(dyn_lds, d.num_pdfs_lds, d.lds_table_cap);
  const int tb = sh.cur_tb, n = sh.cur_n;   // == c.tok_off[frame], c.tok_off[frame + 1] - tb
  const int n_all = sh.cur_n_all;           // tokens created (> n when the commit dropped those that cannot be expanded)
  const u64 key = sh.best_key;
  const int c_lt = sh.c_lt, c_le = sh.c_le;
  const bool ready = sh.cutoff_ready != 0;
  const float rd_cutoff = sh.nx_cur_cutoff, rd_abeam = sh.nx_adaptive_beam;
  float *cost_cache = L.cost_cache;
  const bool cached = sh.cache_valid != 0 && n <= 3 * BIGCAP;
  LdsBarrier();   // everyone has read the stats before a select may reuse Sh scratch
  float best = INFINITY; int best_state = -1;
  if (n > 0) { best = OrderedToFloat(static_cast<u32>(key >> 32)); best_state = static_cast<int>(key & 0xFFFFFFFFu); }
  float cur_cutoff, adaptive_beam;
  if (ready) {                 // evaluated by the commit that made this list, on every token it created
    cur_cutoff = rd_cutoff; adaptive_beam = rd_abeam;
  } else {
    lds_cfloat *cache_l = (lds_cfloat *)cost_cache;
    // the costs of a list too long for the LDS copy are read where they lie (the lane's arena: one pointer of the view)
    const float *cost = d.tok_cost + d.lane_tok_base[Opaque(lane)] + tb;
    GetCutoff(cfg, n, best, c_lt, c_le,
              [&]() { return cached ? BlockSelectLinear(cache_l, n, cfg.max_active, best, cfg.beam, L.lh, L.cand, &sh)
                                    : BlockSelectLinear(cost, n, cfg.max_active, best, cfg.beam, L.lh, L.cand, &sh); },
              [&]() { return cached ? BlockSelectKth(cache_l, n, cfg.min_active, &sh) : BlockSelectKth(cost, n, cfg.min_active, &sh); },
              &cur_cutoff, &adaptive_beam);
  }
  LdsBarrier();
  Stamp(&sh, PH_CUTOFF);
  fc.tb = tb; fc.n = n; fc.n_all = n_all; fc.best = best; fc.best_state = best_state; fc.cur_cutoff = cur_cutoff; fc.adaptive_beam = adaptive_beam;
  // search mode 2: the seed bound replaces the final one on the frames where the reference's order-dependent extras
  // can matter, i.e. where max_active / min_active made the adaptive beam differ from the beam; with adaptive_beam ==
  // beam the next frame's cutoff (best + beam) equals this frame's final bound and every extra is dead on arrival
  fc.loose = d.loose != 0 && adaptive_beam != cfg.beam;
  fc.big = L.cap_big > L.cap_small && n_all > d.big_frame_tokens;      // (token counts move slowly from frame to frame)
}

// ---- cost offset + seed of next_cutoff from the best token's arcs (:757-772), then ProcessEmitting (:783-815).
// Tokens with <= SMALL_DEG arcs are expanded by their own thread; the rest (LM hubs, trie fan-outs) are queued and
// expanded by groups of lanes / by the whole workgroup.
__device__ __forceinline__ void PhaseExpand(int lane, Sh &sh, unsigned char *dyn_lds, int frame, const float *ll, FrameCtl &fc) {
  const DecDev d = LoadDecDev();
  const Ctx c = MakeCtx(d, Opaque(lane));
  const AdvLds L = MakeAdvLds(dyn_lds, d.num_pdfs_lds, d.lds_table_cap);
  const Tbl tbl = FrameTable(L, fc.big, d.hash_mask);       // (the expansion only records candidates: no table access)
  int2 *big_ta = L.big_ta; int *big_scan = L.big_scan;
  const int tid = Tid();
  const int tb = fc.tb, n = fc.n;
  const float best = fc.best; const int best_state = fc.best_state;
  const float cur_cutoff = fc.cur_cutoff, adaptive_beam = fc.adaptive_beam;
  const bool loose = fc.loose;
  const float *cost = c.tok_cost + tb;
  const int *state = c.tok_state + tb;
  const float cost_offset = (n > 0) ? -best : 0.0f;
  // the frame's log-likelihood row is in LDS once every wavefront's DMA of it has landed (issued a frame ago)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    c.cost_offsets[frame] = cost_offset;
    c.trace_ntok[frame] = fc.n_all; c.trace_cutoff[frame] = cur_cutoff;
    c.lnk_off[2 * (frame + 1)] = sh.lnk_used;
    sh.next_cutoff_u = FloatToOrdered(INFINITY);       // (the last frame's value was handed on as a parameter)
  }
  LlRow row; row.g = ll; row.l = (lds_cfloat *)L.ll; row.n_lds = d.num_pdfs_lds;
  LdsBarrier();
  if (n > 0 && static_cast<u32>(best_state) >= static_cast<u32>(d.g.num_states)) {
    if (tid == 0) KAMD_OOB_PRINTF("KAMD-OOB site=5 lane=%d frame=%d best_state=%d n=%d n_all=%d tb=%d\n", (int)blockIdx.x, frame, best_state, n, fc.n_all, tb);
    atomicOr(&sh.err, ERR_BAD_STATE(5));
  } else if (n > 0) {
    const u32 a0 = d.g.off[best_state].x, a1 = d.g.off[best_state + 1].x;
    float seed = INFINITY;
    for (u32 a = a0 + tid; a < a1; a += NT) {
      const uint2 hot = d.e_hot[a];
      const float new_weight = __uint_as_float(hot.x) + cost_offset - LogLikePdf(row, static_cast<int>(hot.y & ~EPS_FLAG)) + best;
      seed = fminf(seed, new_weight + adaptive_beam);
    }
    // a minimum is a minimum in any order: the wavefronts that hold an arc of the best token put theirs straight
    // onto the running bound the expansion tightens further (usually one wavefront: no workgroup reduction)
    seed = WaveMinF(seed);
    if ((tid & 63) == 0 && seed < INFINITY) atomicMin(&sh.next_cutoff_u, FloatToOrdered(seed));
  }
  LdsBarrier();
  Stamp(&sh, PH_SEED);
  const float seed_cutoff = OrderedToFloat(sh.next_cutoff_u);   // :757-772, before any other arc tightens it
  const int link_base = sh.lnk_used;
  int n_exp = 0; long long a_emit = 0;
  // EXPT tokens per thread per outer iteration: their costs, states and arc ranges are all
  // loaded before the first one is expanded (three dependent round trips per BATCH of
  // tokens instead of per token).  EXPT * NT = BIGCAP tokens are queued at most, so the
  // queue is flushed after every outer iteration and cannot overflow.
  // "good tokens first" (measured and left off): with d.good_first > 0 the tokens within that
  // distance of the best one are expanded in a pass of their own before everybody else, so that the running bound the
  // others are filtered with is close to its final value (fewer candidates recorded); costs the token list a second read.
  const int n_pass = d.good_first > 0.0f ? 2 : 1;
  const float good_cut = best + d.good_first;
  for (int pass = 0; pass < n_pass; pass++)
  for (int base = 0; base < n; base += EXPT * NT) {
    float tcost[EXPT]; int tstate[EXPT]; u32 ta0[EXPT], ta1[EXPT];
#pragma unroll
    for (int k = 0; k < EXPT; k++) {
      const int i = base + tid + k * NT;
      // (unconditional loads at a clamped index: a load inside a conditional block is waited for at the block's end,
      // which would serialize the EXPT fetches)
      const int ic = min(i, n - 1);
      tcost[k] = cost[ic]; tstate[k] = state[ic];
      if (i >= n) tcost[k] = INFINITY;
      if (static_cast<u32>(tstate[k]) >= static_cast<u32>(d.g.num_states)) {
        if (i < n) {
          KAMD_OOB_PRINTF("KAMD-OOB site=6 lane=%d tid=%d frame=%d state=%d cost=%f i=%d n=%d n_all=%d tb=%d\n", (int)blockIdx.x, tid, frame, tstate[k], tcost[k], i, n, fc.n_all, tb);
          atomicOr(&sh.err, ERR_BAD_STATE(6));
        } else atomicOr(&sh.err, ERR_BAD_STATE(7));
        tstate[k] = 0; tcost[k] = INFINITY;
      }
    }
#pragma unroll
    for (int k = 0; k < EXPT; k++) {
      const uint2 o0 = d.g.off[tstate[k]], o1 = d.g.off[tstate[k] + 1];
      const bool mine = n_pass == 1 || (pass == 0) == (tcost[k] <= good_cut);
      const bool live = base + tid + k * NT < n && tcost[k] <= cur_cutoff && mine;      // :787 (the cutoff may be +inf)
      ta0[k] = live ? o0.x : 0u; ta1[k] = live ? o1.x : 0u;
    }
#pragma unroll
    for (int k = 0; k < EXPT; k++) {
      const int i = base + tid + k * NT;
      if (i < n && tcost[k] <= cur_cutoff && (n_pass == 1 || (pass == 0) == (tcost[k] <= good_cut))) {
        const float cur_cost = tcost[k];
        n_exp++;
        const u32 a0 = ta0[k], a1 = ta1[k];
        const u32 deg = a1 - a0;
        a_emit += deg;
        if (deg <= SMALL_DEG) {
          uint2 hot[SMALL_DEG]; u32 aidx[SMALL_DEG]; int tok[SMALL_DEG]; float cst[SMALL_DEG]; bool ok[SMALL_DEG];
#pragma unroll
          for (int q = 0; q < SMALL_DEG; q++) {
            ok[q] = static_cast<u32>(q) < deg;
            tok[q] = tb + i; cst[q] = cur_cost; aidx[q] = a0 + q;
            hot[q] = d.e_hot[ok[q] ? a0 + q : 0u];
          }
          ProcessArcs<SMALL_DEG>(d, c, &sh, tbl, row, hot, aidx, tok, cst, ok, cost_offset, adaptive_beam, link_base, loose, seed_cutoff);
        } else {
          // one entry per token (at most EXPT * NT = BIGCAP per outer iteration): tokens of up to HUGE_DEG arcs fill the
          // queue from the bottom, the few with more (the LM's start / backoff hubs) from the top
          int p;
          if (deg <= HUGE_DEG) p = WaveAlloc(&sh.bigcnt); else p = BIGCAP - 1 - WaveAlloc(&sh.hugecnt);
          big_ta[p] = make_int2(i, static_cast<int>(a0)); big_scan[p] = deg;
        }
      }
    }
    LdsBarrier();
    const int nb = sh.bigcnt, nh = sh.hugecnt;   // uniform: read between two barriers
    LdsBarrier();
    if (nb + nh > 0) {
      Stamp(&sh, PH_EXPAND);
      // A queued token is expanded by a GROUP of lanes, 4 arcs per lane and trip (all loaded before the first is
      // used): consecutive lanes read consecutive 8-byte records of one state, the token's index and cost are read once
      // per token, and no arc needs a search for its owner (the flattened arc-parallel form this replaces spent more
      // than half of the expansion's issue slots on that search).  Groups take the tokens round robin; the degrees of
      // the states that matter (LM history states: tens of arcs) make the trips of a wavefront's groups alike.
      {
        const int grp = tid / GL, sub = tid % GL;
        // two tokens per group and trip: the records of both are in flight together (a trip is one HBM round trip; what
        // bounds the expansion is how many of them a CU has outstanding)
        for (int e = grp; e < nb; e += TPG * (NT / GL)) {
          int2 ta[TPG]; int deg[TPG]; float cs[TPG];
          int dmax = 0;
#pragma unroll
          for (int t = 0; t < TPG; t++) {
            const int et = e + t * (NT / GL);
            const bool have = et < nb;
            ta[t] = big_ta[have ? et : e];
            deg[t] = have ? big_scan[et] : 0;
            dmax = max(dmax, deg[t]);
          }
#pragma unroll
          for (int t = 0; t < TPG; t++) cs[t] = cost[ta[t].x];
          for (int k0 = 0; k0 < dmax; k0 += 4 * GL) {
            uint2 hot[4 * TPG]; u32 aidx[4 * TPG]; int tok[4 * TPG]; float cst[4 * TPG]; bool ok[4 * TPG];
#pragma unroll
            for (int q = 0; q < 4 * TPG; q++) {
              const int k = k0 + sub + GL * (q & 3);
              ok[q] = k < deg[q >> 2];
              aidx[q] = static_cast<u32>(ta[q >> 2].y) + static_cast<u32>(k);
              tok[q] = tb + ta[q >> 2].x; cst[q] = cs[q >> 2];
              hot[q] = d.e_hot[ok[q] ? aidx[q] : static_cast<u32>(ta[q >> 2].y)];
            }
            ProcessArcs<4 * TPG>(d, c, &sh, tbl, row, hot, aidx, tok, cst, ok, cost_offset, adaptive_beam, link_base, loose, seed_cutoff);
          }
        }
      }
      // the hubs: every thread of the lane on one token's arcs, ARCW per thread per trip
      for (int h = 0; h < nh; h++) {
        const int2 ta = big_ta[BIGCAP - 1 - h];
        const int deg = big_scan[BIGCAP - 1 - h];
        const float cst1 = cost[ta.x];
        for (int k0 = tid; k0 < deg; k0 += ARCW * NT) {
          uint2 hot[ARCW]; u32 aidx[ARCW]; int tok[ARCW]; float cst[ARCW]; bool ok[ARCW];
#pragma unroll
          for (int q = 0; q < ARCW; q++) {
            const int k = k0 + q * NT;
            ok[q] = k < deg;
            aidx[q] = static_cast<u32>(ta.y) + static_cast<u32>(k);
            tok[q] = tb + ta.x; cst[q] = cst1;
            hot[q] = d.e_hot[ok[q] ? aidx[q] : static_cast<u32>(ta.y)];
          }
          ProcessArcs<ARCW>(d, c, &sh, tbl, row, hot, aidx, tok, cst, ok, cost_offset, adaptive_beam, link_base, loose, seed_cutoff);
        }
      }
      LdsBarrier();
      if (tid == 0) { sh.bigcnt = 0; sh.hugecnt = 0; }
      LdsBarrier();
      Stamp(&sh, PH_EXPAND_BIG);
    }
  }
  LdsBarrier();
  Stamp(&sh, PH_EXPAND);
  {
    const int ne = WaveSumI(n_exp), ae = WaveSumI(static_cast<int>(a_emit));
    if ((tid & 63) == 0) {   // work counters: running sums, nobody needs this frame's total
      atomicAdd(reinterpret_cast<unsigned long long *>(&sh.cnt[0]), static_cast<unsigned long long>(ne));
      atomicAdd(reinterpret_cast<unsigned long long *>(&sh.cnt[1]), static_cast<unsigned long long>(ae));
      atomicAdd(reinterpret_cast<unsigned long long *>(&sh.cnt[2]), static_cast<unsigned long long>(ae));
    }
    if (tid == 0) {
      sh.cnt[6] += 1;
      c.lnk_off[2 * (frame + 1) + 1] = link_base + min(sh.n_links, c.lnk_cap - link_base);   // for finalize / the host
    }
  }
  fc.cost_offset = cost_offset; fc.seed_cutoff = seed_cutoff; fc.link_base = link_base;
  __syncthreads();   // FULL barrier: InsertEmitted reads the links other threads recorded (global)
  fc.next_cutoff = OrderedToFloat(sh.next_cutoff_u);
}

// ---- FindOrAddToken for the recorded links, against the final cutoff.  `may_preselect`: the tokens this frame creates
// will be looked at by nobody but the next frame of this very call (a work-queue lane, not the call's last frame).
__device__ __forceinline__ void PhaseInsert(int lane, Sh &sh, unsigned char *dyn_lds, FrameCtl &fc, bool may_preselect) {
  const DecDev d = LoadDecDev();
  const Ctx c = MakeCtx(d, Opaque(lane));
  const AdvLds L = MakeAdvLds(dyn_lds, d.num_pdfs_lds, d.lds_table_cap);
  const int tid = Tid();
  const float cutoff = fc.loose ? fc.seed_cutoff : fc.next_cutoff;
  const int n_cand = max(min(sh.n_links, c.lnk_cap - fc.link_base), 0);
  fc.hmask = d.full_level2 ? d.hash_mask : FrameLevel2Mask(sh.n_links, d.hash_cap);
  fc.presel = false; fc.presel_bound = INFINITY;
  // ---- pre-selection (see InsertEmitted): pick the bound from a histogram of the candidates' costs
  // candidates wanted under the bound: max_active plus a margin for the candidates that are second arcs into a state (they
  // create no token).  Too small a margin and the count of created tokens falls short -- the bound is raised and another
  // slab of candidates inserted, a sweep more; too large and tokens are created for nothing (measured at the planted
  // load with a fixed margin: 5 % 358 ms, 25 % 372 ms, 50 % 390 ms; 1 % 450 ms when a shortfall still meant inserting
  // everything; the random-score load has more duplicates: 8 % 262 ms, 15 % 249 ms).  The lane starts at ps_margin_pct and
  // follows what its frames show: the candidates per created token of the last proven bound, half as much again, + 2 %.
  const int margin_pm = (sh.ps_margin_pm >= 0 && d.ps_adapt) ? sh.ps_margin_pm : 10 * d.ps_margin_pct;
  const int ps_target = d.cfg.max_active + max(static_cast<int>(static_cast<long long>(d.cfg.max_active) * margin_pm / 1000), 16);
  float lo = 0.0f, scale = 0.0f;
  int bin = -1, under = 0, total = 0;
  // the first bucket of the candidates' cost histogram (Sh::hist) under whose upper edge `target` candidates lie, and how many
  // lie under it; SH_HIST / NT buckets per thread, workgroup scan.  Uniform; ends on a barrier.
  auto find_bin = [&](int target, int *bin_out, int *under_out, int *total_out) {
    constexpr int PER = SH_HIST / NT;
    int h[PER], mine = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { h[q] = static_cast<int>(sh.hist[PER * tid + q]); mine += h[q]; }
    const int incl = WaveInclScanI(mine);
    LdsBarrier();                                // (whoever read redi / sel_bin last is done)
    if ((tid & 63) == 63) sh.redi[tid >> 6] = incl;
    if (tid == 0) { sh.sel_bin = -1; sh.sel_below = 0; }
    LdsBarrier();
    int wbase = 0, tot_all = 0;
    for (int q = 0; q < NWAVES; q++) { const int v = sh.redi[q]; if (q < (tid >> 6)) wbase += v; tot_all += v; }
    const int excl = wbase + incl - mine;
    if (target - 1 >= excl && target - 1 < excl + mine) {
      int cum = excl, b = PER * tid;
#pragma unroll
      for (int q = 0; q < PER; q++) { if (cum + h[q] > target - 1) break; cum += h[q]; b++; }
      sh.sel_bin = b; sh.sel_below = cum + static_cast<int>(sh.hist[b]);     // candidates up to and including bucket b
    }
    LdsBarrier();
    *bin_out = sh.sel_bin; *under_out = sh.sel_below; *total_out = tot_all;
    LdsBarrier();
  };
  if (may_preselect && d.preselect != 0 && d.cfg.max_active < (1 << 28) && d.cfg.max_active >= d.cfg.min_active && static_cast<long long>(n_cand) * 100 >= static_cast<long long>(ps_target) * d.ps_worth_pct) {
    lo = fc.next_cutoff - fc.adaptive_beam;         // (about) the best candidate's cost
    const float width = cutoff - lo;
    scale = static_cast<float>(SH_HIST) / width;
    if (width > 0.0f && scale < 3.0e38f) {      // uniform
      for (int i = tid; i < SH_HIST; i += NT) sh.hist[i] = 0;
      LdsBarrier();
      const uint4 *cand = CandBase(c, fc.link_base);
      for (int i0 = tid; i0 < n_cand; i0 += INSB * NT) {
        float tot[INSB];
#pragma unroll
        for (int k = 0; k < INSB; k++) tot[k] = __uint_as_float(cand[min(i0 + k * NT, n_cand - 1)].z);
#pragma unroll
        for (int k = 0; k < INSB; k++)
          if (i0 + k * NT < n_cand && tot[k] <= cutoff) {
            const float b = (tot[k] - lo) * scale;
            atomicAdd(&sh.hist[b < 0.0f ? 0 : min(static_cast<int>(b), SH_HIST - 1)], 1u);
          }
      }
      LdsBarrier();
      find_bin(ps_target, &bin, &under, &total);
      // worth it only when the bound leaves a good part of the candidates out
      fc.presel = bin >= 0 && bin < SH_HIST - 1 && static_cast<long long>(under) * d.ps_worth_pct <= 100ll * total &&
                  lo + static_cast<float>(bin + 1) / scale < cutoff;
    }
  }
  // a pre-selected frame inserts ~1.1 max_active tokens whatever the last frame held: the whole table region then
  if (fc.presel && !fc.big && L.cap_big > L.cap_small && ps_target > d.big_frame_tokens) fc.big = true;
  Tbl tbl = FrameTable(L, fc.big, fc.hmask);
  tbl.q_on = true; tbl.q_lds = reinterpret_cast<u32 *>(L.cost_cache); tbl.q_cap = fc.big ? 0 : (3 * BIGCAP) / 2;    // = PhaseCommit's wl0
  if (fc.big) {   // the upper half of the region was the expansion's queue: make it table
    for (int sl = L.cap_small + tid; sl < L.cap_big; sl += NT) L.T[sl] = EMPTY64;
    LdsBarrier();
  }
  if (fc.presel) {
    // slab by slab until the tokens created under the bound outnumber max_active (usually the first does it); three at most
    float b_prev = -INFINITY;
    int made_total = 0;
    bool proven = false;
    fc.k_surv = 0;
    for (int iter = 0; iter < 3; iter++) {
      const float b_new = lo + static_cast<float>(bin + 1) / scale;
      int made = 0, dummy = 0;
      fc.k_surv += InsertEmitted(d, c, &sh, tbl, fc.link_base, sh.n_links, cutoff, b_prev, b_new, &made);
      BlockSum2<true>(made, dummy, &sh);
      made_total += made;
      b_prev = b_new;
      if (made_total > d.cfg.max_active) { proven = true; break; }     // uniform
      if (iter == 2) break;
      // the next slab: the tokens still missing at the rate seen so far, half as many again
      const long long miss = d.cfg.max_active + 1 - made_total;
      const int more = static_cast<int>(min(miss * 3 * max(under, 1) / (2 * max(made_total, 1)) + 64, 1ll << 28));
      int bin2, under2, total2;
      find_bin(under + more, &bin2, &under2, &total2);
      if (bin2 <= bin || bin2 >= SH_HIST - 1 || static_cast<long long>(under2) * d.ps_worth_pct > 100ll * total2 ||
          !(lo + static_cast<float>(bin2 + 1) / scale < cutoff)) break;
      bin = bin2; under = under2;
    }
    if (!proven) {     // the bounds proved nothing -- insert the rest, the frame is an ordinary one
      int unused = 0;
      fc.k_surv += InsertEmitted(d, c, &sh, tbl, fc.link_base, sh.n_links, cutoff, b_prev, INFINITY, &unused);
      fc.presel = false;
      if (tid == 0) sh.ps_margin_pm = min(2 * margin_pm + 50, 1000);
    } else {
      fc.presel_bound = b_prev;
      if (tid == 0) {
        sh.presel_frames++;
        // candidates per created token under the proven bound, as a margin over max_active (per mille), half as much again + 2 %
        const long long need_pm = 1000ll * max(under, 1) / max(made_total, 1) - 1000;
        const int next_pm = static_cast<int>(min(max(need_pm * 3 / 2 + 20, 30ll), 1000ll));
        sh.ps_margin_pm = max(next_pm, margin_pm * 3 / 4);
      }
    }
  } else {
    int unused = 0;
    fc.k_surv = InsertEmitted(d, c, &sh, tbl, fc.link_base, sh.n_links, cutoff, -INFINITY, INFINITY, &unused);
  }
  Stamp(&sh, PH_FIXUP);
}

// ---- ProcessNonemitting(next_cutoff) + commit of token list frame + 1
__device__ __forceinline__ void PhaseCommit(int lane, Sh &sh, unsigned char *dyn_lds, int frame, const FrameCtl &fc, bool drop) {
  const DecDev d = LoadDecDev();
  const Ctx c = MakeCtx(d, Opaque(lane));
  const AdvLds L = MakeAdvLds(dyn_lds, d.num_pdfs_lds, d.lds_table_cap);
  const Tbl tbl = FrameTable(L, fc.big, fc.hmask);
  CommitLds cl;
  if (fc.big) {        // no idle LDS: the worklists, the owner list and the costs stay in the lane's HBM areas
    cl.wl0 = cl.wl1 = NULL; cl.wl_cap = 0; cl.owners = NULL; cl.owners_cap = 0; cl.cost_cache = NULL; cl.cache_cap = 0;
  } else {
    cl.wl0 = reinterpret_cast<u32 *>(L.cost_cache); cl.wl1 = cl.wl0 + (3 * BIGCAP) / 2; cl.wl_cap = (3 * BIGCAP) / 2;
    cl.owners = reinterpret_cast<uint2 *>(L.lh); cl.owners_cap = (LHBINS + LHCAND > CHUNKCAP ? LHBINS + LHCAND : CHUNKCAP) / 2;
    cl.cost_cache = L.cost_cache; cl.cache_cap = 3 * BIGCAP;
  }
  CommitFrame2(d, c, &sh, tbl, fc.next_cutoff, frame + 1, fc.link_base, fc.k_surv, cl, fc.loose, drop, fc.presel,
               fc.loose ? fc.seed_cutoff : fc.next_cutoff, fc.presel_bound);
}

// AdvanceDecoding (lattice-faster-decoder.cc:593-632): the frame loop of one lane.  Every phase takes its own view of
// the descriptors (LoadDecDev above); what the phases hand to each other is the FrameCtl.
template <bool kDropDead>
__device__ __forceinline__ void AdvanceLane(int lane_in, Sh *shp, unsigned char *dyn_lds, const kamd_decode_task &task) {
  Sh &sh = *shp;
  const int lane = Opaque(lane_in);
  const int tid = Tid();
  int frame;
  {
    const DecDev d = LoadDecDev();
    const Ctx c = MakeCtx(d, Opaque(lane));
    const AdvLds L = MakeAdvLds(dyn_lds, d.num_pdfs_lds, d.lds_table_cap);
    for (int i = Tid(); i < L.cap_small; i += NT) L.T[i] = EMPTY64;     // (the upper half is cleared by the frames that use it)
    InitSh(&sh);
    LaneState *S = c.st;
    frame = S->frame;
    if (S->error || S->finalized) return;
    if (tid == 0) {
      sh.cur_tb = c.tok_off[frame]; sh.cur_n = c.tok_off[frame + 1] - c.tok_off[frame];
      sh.lnk_used = S->lnk_used; sh.round = S->round;
    }
    if (task.n_frames > 0) RowDma(L.ll, d.num_pdfs_lds, task.d_loglikes);
    ComputeFrameStats(d, c, &sh, frame, L.cost_cache, 3 * BIGCAP);
  }
  for (int it = 0; it < task.n_frames; it++, frame++) {
    const float *ll = task.d_loglikes + static_cast<size_t>(it) * task.ld;
    {
      const DecDev d = LoadDecDev();
      if (frame >= d.max_frames) { if (tid == 0) atomicOr(&sh.err, ERR_FRAMES); __syncthreads(); break; }
    }
    FrameCtl fc;
    PhaseCutoff(lane, sh, dyn_lds, fc);
    PhaseExpand(lane, sh, dyn_lds, frame, ll, fc);
    PhaseInsert(lane, sh, dyn_lds, fc, kDropDead && it + 1 < task.n_frames);
    // nobody reads this frame's row any more.  (Issued here and not before the inserts above: those wait for L2 hits,
    // and loads return in order -- behind a cold 24 KB read they took 2.5 us longer; the closure's first loads are
    // cold graph reads themselves.)
    if (it + 1 < task.n_frames) {
      const DecDev d = LoadDecDev();
      const AdvLds L = MakeAdvLds(dyn_lds, d.num_pdfs_lds, d.lds_table_cap);
      RowDma(L.ll, d.num_pdfs_lds, ll + task.ld);
    }
    // kDropDead (the work-queue lane: nothing reads its token lists before FinalizeDecoding): tokens that cannot be expanded
    // get no record -- never on the call's last frame, whose tokens all stay
    PhaseCommit(lane, sh, dyn_lds, frame, fc, kDropDead && it + 1 < task.n_frames);
    const int err_now = sh.err;    // CommitFrame ends with a barrier; nobody writes err before the next one
    LdsBarrier();
    if (err_now) { frame++; break; }
  }
  DrainStores();     // an early exit leaves the next row's DMA in flight: it must have landed before the LDS changes hands
  {
    const DecDev d = LoadDecDev();
    const Ctx c = MakeCtx(d, Opaque(lane));
    PublishLaneEnd(d, c, &sh, frame);
  }
}
KAMD_SEARCH_KERNEL void AdvanceKernel(DecDev d_unused, const kamd_decode_task *tasks) {
  __shared__ Sh sh;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  const kamd_decode_task task = tasks[blockIdx.x];
  AdvanceLane<false>(task.lane, &sh, dyn_lds, task);     // (its token lists can be read between calls: GetRawLattice of a live decoder)
}

// FinalizeDecoding (lattice-faster-decoder.cc:638-653) = PruneForwardLinksFinal (:389-471)
// + PruneForwardLinks(f, delta = 0) for every earlier frame (:312-383), iterated to the
// exact fixpoint, + PruneTokensForFrame; then in-place, order preserving compaction of
// the surviving tokens / links so the host copies only the raw lattice.
__global__ __launch_bounds__(NT, 4) void FinalizeKernel(DecDev d, const int *lanes) {
  __shared__ Sh sh;
  const int lane = lanes[blockIdx.x];
  const Ctx c = MakeCtx(d, lane);
  const int tid = Tid();
  InitSh(&sh);
  LaneState *S = c.st;
  if (S->error || S->finalized) return;
  const int F = S->frame;
  const float lattice_beam = d.cfg.lattice_beam;
  float best_cost, best_with_final;
  FinalCosts(d, c, &sh, F, &best_cost, &best_with_final);
  const bool finals_empty = best_with_final == INFINITY;                // final_costs_.empty()
  const float final_best = finals_empty ? best_cost : best_with_final;  // :583-588
  u32 *bo = reinterpret_cast<u32 *>(c.scratch);                 // ordered keys: base + emitting
  u32 *xo = reinterpret_cast<u32 *>(c.scratch) + d.hash_cap;    // ordered keys: Jacobi target
  // LDS working set of the sweep (frames of up to FIN_CAP tokens): extra costs / forward
  // costs of frame f and of frame f+1 plus the two ordered-key accumulators.
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_lds[];
  u32 *l_bo = reinterpret_cast<u32 *>(fin_lds);
  u32 *l_xo = l_bo + FIN_CAP;
  float *l_x[2] = {reinterpret_cast<float *>(l_xo + FIN_CAP), reinterpret_cast<float *>(l_xo + 2 * FIN_CAP)};
  float *l_c[2] = {reinterpret_cast<float *>(l_xo + 3 * FIN_CAP), reinterpret_cast<float *>(l_xo + 4 * FIN_CAP)};
  bool next_in_lds = false;   // frame f+1's extras / costs are in l_x[1-cur], l_c[1-cur]
  int cur = 0;
  for (int f = F; f >= 0; f--) {
    const int tb = c.tok_off[f], nt = c.tok_off[f + 1] - tb;
    if (nt > d.hash_cap) { if (tid == 0) atomicOr(&sh.err, ERR_INTERNAL); __syncthreads(); break; }
    float *xcur = c.tok_extra + tb;   // extra_cost of frame f (frame f+1 is final already)
    const int eb = c.lnk_off[2 * f + 1], ee = c.lnk_off[2 * f + 2];
    const bool in_lds = nt <= FIN_CAP && (f == F || next_in_lds);
    if (in_lds) {
      // ================= LDS path: no L2 atomics, no cold gathers of frame f+1 =========
      float *lx = l_x[cur], *lc = l_c[cur], *nx = l_x[cur ^ 1], *nc = l_c[cur ^ 1];
      const int tbn = c.tok_off[f + 1];
      for (int i = tid; i < nt; i += NT) {
        const float co = c.tok_cost[tb + i];
        lc[i] = co;
        float b = INFINITY;
        if (f == F) {
          const float fc = finals_empty ? 0.0f : d.g.final[c.tok_state[tb + i]];
          b = co + fc - final_best;
        }
        l_bo[i] = FloatToOrdered(b);
      }
      __syncthreads();
      if (f < F) {
        const int lb = c.lnk_off[2 * (f + 1)], le = c.lnk_off[2 * (f + 1) + 1];
        for (int li = lb + tid; li < le; li += NT) {
          const Link L = c.links[li];
          if (L.dst < 0) { c.links[li].src = -1; continue; }   // dropped by the exact cutoff
          float lec = nx[L.dst - tbn] + ((lc[L.src - tb] + L.ac + L.graph) - nc[L.dst - tbn]);
          if (lec > lattice_beam) { c.links[li].src = -1; continue; }   // excise (:352)
          if (lec < 0.0f) lec = 0.0f;                                    // :360-364
          atomicMin(&l_bo[L.src - tb], FloatToOrdered(lec));
        }
      }
      __syncthreads();
      for (int i = tid; i < nt; i += NT) {
        const u32 b = l_bo[i];
        float v = OrderedToFloat(b);
        if (f == F && v > lattice_beam) v = INFINITY;   // :462-463
        lx[i] = v;
        l_xo[i] = b;
      }
      __syncthreads();
      if (ee > eb) {
        for (int iter = 0; iter < 20000; iter++) {
          for (int li = eb + tid; li < ee; li += NT) {
            const Link L = c.links[li];
            float lec = lx[L.dst - tb] + ((lc[L.src - tb] + L.ac + L.graph) - lc[L.dst - tb]);
            if (lec > lattice_beam) continue;
            if (lec < 0.0f) lec = 0.0f;
            atomicMin(&l_xo[L.src - tb], FloatToOrdered(lec));
          }
          __syncthreads();
          int changed = 0;
          for (int i = tid; i < nt; i += NT) {
            float v = OrderedToFloat(l_xo[i]);
            if (f == F && v > lattice_beam) v = INFINITY;
            if (!(v == lx[i])) changed = 1;
            lx[i] = v;
            l_xo[i] = l_bo[i];   // re-arm
          }
          if (!__syncthreads_or(changed)) break;
        }
        for (int li = eb + tid; li < ee; li += NT) {
          const Link L = c.links[li];
          const float lec = lx[L.dst - tb] + ((lc[L.src - tb] + L.ac + L.graph) - lc[L.dst - tb]);
          if (lec > lattice_beam) c.links[li].src = -1;
        }
      }
      for (int i = tid; i < nt; i += NT) xcur[i] = lx[i];   // the compaction reads extra_cost from HBM
      __syncthreads();
      next_in_lds = true;
      cur ^= 1;
      continue;
    }
    next_in_lds = false;
    // ================= HBM path (frames larger than FIN_CAP tokens) ======================
    // base term: final-cost term on the last frame (:430), +inf elsewhere (:341)
    for (int i = tid; i < nt; i += NT) {
      float b = INFINITY;
      if (f == F) {
        const float fc = finals_empty ? 0.0f : d.g.final[c.tok_state[tb + i]];
        b = c.tok_cost[tb + i] + fc - final_best;
      }
      __hip_atomic_store(&bo[i], FloatToOrdered(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    DrainStores();
    __syncthreads();
    // emitting links out of frame f (written at step f+1)
    if (f < F) {
      const int lb = c.lnk_off[2 * (f + 1)], le = c.lnk_off[2 * (f + 1) + 1];
      for (int li = lb + tid; li < le; li += NT) {
        const Link L = c.links[li];
        if (L.dst < 0) { c.links[li].src = -1; continue; }   // dropped by the exact cutoff
        float lec = c.tok_extra[L.dst] + ((c.tok_cost[L.src] + L.ac + L.graph) - c.tok_cost[L.dst]);
        if (lec > lattice_beam) { c.links[li].src = -1; continue; }   // excise (:352)
        if (lec < 0.0f) lec = 0.0f;                                    // :360-364
        atomicMin(&bo[L.src - tb], FloatToOrdered(lec));
      }
    }
    __syncthreads();
    for (int i = tid; i < nt; i += NT) {
      const u32 b = LoadU32(&bo[i]);
      float v = OrderedToFloat(b);
      if (f == F && v > lattice_beam) v = INFINITY;   // :462-463
      xcur[i] = v;
      if (ee > eb) __hip_atomic_store(&xo[i], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    DrainStores();
    __syncthreads();
    // epsilon links inside frame f: Jacobi iteration to the exact fixpoint (the
    // reference's "while (changed)" loop; epsilon links are acyclic so it is unique)
    if (ee > eb) {
      for (int iter = 0; iter < 20000; iter++) {
        for (int li = eb + tid; li < ee; li += NT) {
          const Link L = c.links[li];
          float lec = xcur[L.dst - tb] + ((c.tok_cost[L.src] + L.ac + L.graph) - c.tok_cost[L.dst]);
          if (lec > lattice_beam) continue;
          if (lec < 0.0f) lec = 0.0f;
          atomicMin(&xo[L.src - tb], FloatToOrdered(lec));
        }
        __syncthreads();
        int changed = 0, dummy = 0;
        for (int i = tid; i < nt; i += NT) {
          float v = OrderedToFloat(LoadU32(&xo[i]));
          if (f == F && v > lattice_beam) v = INFINITY;
          if (!(v == xcur[i])) changed = 1;
          xcur[i] = v;
          __hip_atomic_store(&xo[i], LoadU32(&bo[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
        }
        DrainStores();
        BlockSum2(changed, dummy, &sh);
        if (changed == 0) break;
      }
      for (int li = eb + tid; li < ee; li += NT) {
        const Link L = c.links[li];
        const float lec = xcur[L.dst - tb] + ((c.tok_cost[L.src] + L.ac + L.graph) - c.tok_cost[L.dst]);
        if (lec > lattice_beam) c.links[li].src = -1;
      }
    }
    __syncthreads();
  }
  Stamp(&sh, PH_FIN_SWEEP);
  // ---- PruneTokensForFrame + GetRawLattice staging: in-place compaction toward the front,
  // frames stay contiguous and in order (order inside a frame is arbitrary: the host sorts by
  // HCLG state).  Per 1024-chunk: read into registers -> barrier -> wavefront-ballot
  // allocated writes; a write never lands beyond the chunk that has already been read.
  int *new_off = reinterpret_cast<int *>(c.wl0);   // [F+2] (hash_cap >= max_frames+2 is checked on the host)
  if (tid == 0) sh.n_new = 0;
  __syncthreads();
  for (int f = 0; f <= F; f++) {
    const int tb = c.tok_off[f], te = c.tok_off[f + 1];
    if (tid == 0) new_off[f] = sh.n_new;     // ordered by the barriers of the previous chunk
    for (int base = tb; base < te; base += NT) {
      const int i = base + tid;
      bool keep = false; int st = 0; float co = 0.f;
      if (i < te) { keep = c.tok_extra[i] != INFINITY; st = c.tok_state[i]; co = c.tok_cost[i]; }
      __syncthreads();
      int pos = -1;
      if (keep) { pos = WaveAlloc(&sh.n_new); c.tok_state[pos] = st; c.tok_cost[pos] = co; }
      if (i < te) c.tok_map[i] = pos;
    }
    __syncthreads();
  }
  const int n_out_tok = sh.n_new;
  if (tid == 0) new_off[F + 1] = n_out_tok;
  __syncthreads();
  // ---- links: drop excised, remap endpoints, remove cost offsets (GetRawLattice :173-180)
  if (tid == 0) sh.n_links = 0;
  __syncthreads();
  for (int s = 0; s <= F; s++) {
    for (int part = 0; part < 2; part++) {
      const int lb = c.lnk_off[2 * s + part], le = c.lnk_off[2 * s + part + 1];
      const float off = (part == 0 && s > 0) ? c.cost_offsets[s - 1] : 0.0f;
      for (int base = lb; base < le; base += NT) {
        const int li = base + tid;
        bool keep = false; Link L;
        L.src = L.dst = L.ilabel = L.olabel = 0; L.graph = L.ac = 0.f;
        if (li < le) {
          L = c.links[li];
          keep = L.src >= 0 && L.dst >= 0;
          if (keep) {
            const int ms = c.tok_map[L.src], md = c.tok_map[L.dst];
            keep = ms >= 0 && md >= 0;
            L.src = ms; L.dst = md;
            if (part == 0) L.ac = L.ac - off;
          }
        }
        __syncthreads();
        if (keep) c.links[WaveAlloc(&sh.n_links)] = L;
      }
    }
  }
  __syncthreads();
  const int n_out_link = sh.n_links;
  for (int f = tid; f <= F + 1; f += NT) c.tok_off[f] = new_off[f];
  Stamp(&sh, PH_FIN_COMPACT);
  if (tid == 0) {
    for (int i = 0; i < 16; i++) S->phase_cycles[i] += sh.ph[i];
    S->finalized = 1;
    S->final_best_cost = final_best;
    S->final_relative_cost = (best_cost == INFINITY && best_with_final == INFINITY) ? INFINITY : best_with_final - best_cost;
    S->out_ntok = n_out_tok; S->out_nlink = n_out_link;
    S->out_tok_base = 0; S->out_lnk_base = 0; S->out_cost_in_map = 0;
    S->error |= sh.err;
  }
}

// one slot from a workgroup counter that counts DOWN (staging areas grow from the top of the
// arena toward the data still to be swept); call under the predicate.
__device__ inline int WaveAllocDown(int *top) {
  const u64 m = __ballot(1);
  const int lane = Tid() & 63;
  const int leader = __ffsll(static_cast<long long>(m)) - 1;
  int base = 0;
  if (lane == leader) base = atomicSub(top, __popcll(m));
  base = __builtin_amdgcn_readfirstlane(base);
  return base - 1 - __popcll(m & ((1ull << lane) - 1ull));
}

// what the sweep needs of a Link, with the endpoints as frame-local 16-bit indices
// (src - first token of frame f) | (dst - first token of dst's frame) << 16; 0xFFFF = dropped
struct LinkLite { u32 sd; float graph, ac; };
__device__ inline LinkLite Lite(const Link &L, int src_base, int dst_base) {
  LinkLite r;
  r.sd = static_cast<u32>(L.src - src_base) & 0xFFFFu;
  r.sd |= (L.dst < 0 ? 0xFFFFu : (static_cast<u32>(L.dst - dst_base) & 0xFFFFu)) << 16;
  r.graph = L.graph; r.ac = L.ac;
  return r;
}
// FIN_PF / FIN_TR (template parameters of the sweep): links / tokens per thread held in registers one frame ahead; the rest
// of a larger frame is read in place.  The standalone kernels keep a whole LDS-mode frame (FIN_CAP / NT tokens, 4 links
// per thread: 52 VGPRs that fit their own 128), the fused work-queue kernel two of each (22 VGPRs: with more, the compiler
// spilled the pipeline and every "prefetched" record was waited for at once, to be stored to scratch).
#define FIN_W (NT / 2)                    // frames whose offsets the sweep keeps in LDS
#define FIN_LDS_BYTES (6 * FIN_CAP * 4 + (4 * FIN_W + 11) * 4)

// FinalizeDecoding, second generation: the same exact backward sweep as FinalizeKernel, with
//  * the surviving tokens / links of frame f emitted to a staging area as soon as frame f is
//    final (no marking pass, no separate compaction sweeps over the whole arena).  Staging
//    grows DOWN from the top of the lane's arenas: after frames F..f it holds at most as many
//    records as those frames held, so it never reaches data that is still to be swept;
//  * frame f-1's token costs / states and links loaded into registers while frame f is being
//    processed (the sweep is a chain of dependent per-frame steps: latency, not bandwidth);
//  * the epsilon fixpoint iterated in place on one ordered-key array (values only decrease,
//    so chaotic relaxation reaches the same unique fixpoint as the Jacobi form).
// Output: tokens in tok_state / tok_map (cost bits) at [out_tok_base, tok_cap), frame by
// frame; links at [out_lnk_base, lnk_cap) with src/dst = arena positions of the tokens.
struct FinSh { int tok_top, lnk_top, chg[3]; };
// MID = true is PruneActiveTokens in the middle of an utterance (lattice-faster-decoder.cc:519-546, with the exact
// fixpoint instead of the delta tolerance): every token of the newest frame keeps extra cost 0 (PruneForwardLinks
// :312-383), the survivors are staged exactly as above and then moved back to the bottom of the arenas with the per-frame
// offsets rebuilt, so that AdvanceDecoding continues on a lane whose dead tokens and links are gone.  Nothing that the
// final sweep keeps is ever dropped here: a token's extra cost against the current frontier is a lower bound of its
// final one, and the final sweep's minima are over links this one keeps.
template <bool MID, int FIN_TR, int FIN_PF>
__device__ __forceinline__ void FinalizeLane2(const DecDev &d, const Ctx &c, Sh *shp, FinSh *fs, unsigned char *fin_lds) {
  Sh &sh = *shp;
  int &s_tok_top = fs->tok_top, &s_lnk_top = fs->lnk_top;
  int *s_chg = fs->chg;
  const int tid = Tid();
  InitSh(&sh);
  LaneState *S = c.st;
  if (S->error || S->finalized) return;
  const int F = S->frame;
  const float lattice_beam = d.cfg.lattice_beam;
  float best_cost = 0.0f, best_with_final = 0.0f;
  if (!MID) FinalCosts(d, c, &sh, F, &best_cost, &best_with_final);
  const bool finals_empty = best_with_final == INFINITY;                // final_costs_.empty()
  const float final_best = finals_empty ? best_cost : best_with_final;  // :583-588
  // MID: where the rebuilt link offsets wait until the move (the tail of the lane's scratch, behind bo / xo)
  int *nlo = reinterpret_cast<int *>(c.scratch) + 2 * d.hash_cap;
  if (MID && 2 * F + 3 > 2 * LDS_TABLE_CAP) { if (tid == 0) atomicOr(&S->error, ERR_FRAMES); return; }
  const u32 INF_O = FloatToOrdered(INFINITY);
  u32 *bo = reinterpret_cast<u32 *>(c.scratch);                 // HBM mode: base + emitting
  u32 *xo = reinterpret_cast<u32 *>(c.scratch) + d.hash_cap;    // HBM mode: Jacobi target
  // HBM mode: frame-local token -> staged position.  (Not c.stamp: the epsilon-closure stamps must survive into the
  // lane's next utterance, a stale position equal to a later round number would suppress a re-queue.)
  int *gpos[2] = {c.slot_tok, reinterpret_cast<int *>(c.wl1)};
  int *new_off = reinterpret_cast<int *>(c.wl0);                 // [F+2] staged start of every frame
  float *stage_cost = reinterpret_cast<float *>(c.tok_map);
  // LDS: [2] ordered extra costs, [2] forward costs, [2] staged positions (frame f / frame f+1)
  u32 *l_base = reinterpret_cast<u32 *>(fin_lds);
  // The per-frame offsets of the next FIN_W frames, in LDS: the sweep is a chain of dependent steps per frame and a frame's
  // records cannot be requested before its offsets are known -- read from global (written up to a thousand frames ago: L2
  // misses) they put a second round trip in front of every frame's prefetch.
  int *l_tok = reinterpret_cast<int *>(fin_lds + 6 * FIN_CAP * 4);      // tok_off[wb .. wb + FIN_W + 3)
  int *l_lnk = l_tok + FIN_W + 3;                                        // lnk_off[2 wb .. 2 wb + 2 FIN_W + 6)
  float *l_cof = reinterpret_cast<float *>(l_lnk + 2 * FIN_W + 6);       // cost_offsets[wb .. wb + FIN_W + 2)
  int wb = F > FIN_W ? F - FIN_W : 0;
  auto refill = [&]() {
    for (int i = tid; i < FIN_W + 3; i += NT) l_tok[i] = wb + i <= F + 1 ? c.tok_off[wb + i] : 0;
    for (int i = tid; i < 2 * FIN_W + 6; i += NT) l_lnk[i] = 2 * wb + i <= 2 * F + 2 ? c.lnk_off[2 * wb + i] : 0;
    for (int i = tid; i < FIN_W + 2; i += NT) l_cof[i] = wb + i < F ? c.cost_offsets[wb + i] : 0.0f;
    __syncthreads();
  };
  auto TOF = [&](int x) -> int { return l_tok[x - wb]; };
  auto LOF = [&](int x) -> int { return l_lnk[x - 2 * wb]; };
  refill();
  if (tid == 0) {
    s_tok_top = c.tok_cap; s_lnk_top = c.lnk_cap; new_off[F + 1] = c.tok_cap; s_chg[0] = s_chg[1] = s_chg[2] = 0;
    if (MID) nlo[2 * F + 2] = c.lnk_cap;
  }
  // registers: frame f (cur) and frame f-1 (nxt)
  LinkLite rl[FIN_PF];     // frame f, packed
  int2 nsd[FIN_PF]; float2 nga[FIN_PF];   // frame f-1 as loaded (nothing may be computed on them before the next iteration: that would wait for the loads)
  float rc[FIN_TR], nc_[FIN_TR];
  int rs[FIN_TR], ns_[FIN_TR];
  auto prefetch = [&](int f, int2 *psd, float2 *pga, float *pc, int *ps) {
    if (f < 0) return;
    const int tb = TOF(f), nt = TOF(f + 1) - tb;
    const int eb = LOF(2 * f + 1);
    const int le = f < F ? LOF(2 * (f + 1) + 1) : LOF(2 * f + 2);
#pragma unroll
    for (int k = 0; k < FIN_TR; k++) {
      const int i = tid + k * NT;
      pc[k] = 0.f; ps[k] = 0;
      if (i < nt) { pc[k] = c.tok_cost[tb + i]; ps[k] = c.tok_state[tb + i]; }
    }
#pragma unroll
    for (int k = 0; k < FIN_PF; k++) {
      const int li = eb + tid + k * NT;
      psd[k] = make_int2(0, -1); pga[k] = make_float2(0.f, 0.f);
      if (li < le) {
        psd[k] = *reinterpret_cast<const int2 *>(&c.links[li].src);
        pga[k] = *reinterpret_cast<const float2 *>(&c.links[li].graph);
      }
    }
  };
  auto promote = [&](int f) {   // pack the registers loaded for frame f
    if (f < 0) return;
    const int tb = TOF(f), tbn = TOF(f + 1);
    const int eb = LOF(2 * f + 1), ee = LOF(2 * f + 2);
#pragma unroll
    for (int k = 0; k < FIN_PF; k++) {
      Link L; L.src = nsd[k].x; L.dst = nsd[k].y; L.graph = nga[k].x; L.ac = nga[k].y; L.ilabel = 0; L.olabel = 0;
      rl[k] = Lite(L, tb, (eb + tid + k * NT) < ee ? tb : tbn);
    }
#pragma unroll
    for (int k = 0; k < FIN_TR; k++) { rc[k] = nc_[k]; rs[k] = ns_[k]; }
  };
  prefetch(F, nsd, nga, nc_, ns_);
  promote(F);
  __syncthreads();
  bool prev_lds = false;   // frame f+1 was processed in LDS mode (its x / cost / pos are in LDS)
  int jit = 0;             // epsilon-fixpoint rounds so far (all frames)
  int cur = 0;
  for (int f = F; f >= 0; f--) {
    if (wb > 0 && f - 1 < wb) {      // uniform: slide the window of offsets
      LdsBarrier();                  // promote() of the previous iteration has read the old one
      wb = f > FIN_W ? f - FIN_W : 0;
      refill();
    }
    const int tb = TOF(f), nt = TOF(f + 1) - tb;
    const int tbn = TOF(f + 1), ntn = f < F ? TOF(f + 2) - tbn : 0;
    if (nt > d.hash_cap) { if (tid == 0) atomicOr(&sh.err, ERR_INTERNAL); __syncthreads(); break; }
    const int eb = LOF(2 * f + 1), ee = LOF(2 * f + 2);       // epsilon links inside frame f
    const int le = f < F ? LOF(2 * (f + 1) + 1) : ee;              // emitting links f -> f+1: [ee, le)
    const float emit_off = f < F ? l_cof[f - wb] : 0.0f;                        // GetRawLattice :173-180 (used for f < F only)
    prefetch(f - 1, nsd, nga, nc_, ns_);
    const int lnk_top0 = s_lnk_top;   // uniform: the previous iteration ended with a barrier
    const bool lds_mode = nt <= FIN_CAP && ntn <= FIN_CAP;
    const bool next_hbm_mode = f > 0 && ((TOF(f) - TOF(f - 1)) > FIN_CAP || nt > FIN_CAP);
    int *gp = gpos[f & 1], *gpn = gpos[(f + 1) & 1];
    if (lds_mode) {
      u32 *lx = l_base + cur * FIN_CAP, *nx = l_base + (cur ^ 1) * FIN_CAP;
      float *lc = reinterpret_cast<float *>(l_base + (2 + cur) * FIN_CAP), *nc = reinterpret_cast<float *>(l_base + (2 + (cur ^ 1)) * FIN_CAP);
      int *lp = reinterpret_cast<int *>(l_base + (4 + cur) * FIN_CAP), *np = reinterpret_cast<int *>(l_base + (4 + (cur ^ 1)) * FIN_CAP);
      if (f < F && !prev_lds) {   // frame f+1 went through HBM mode: fetch its results
        for (int i = tid; i < ntn; i += NT) {
          nx[i] = FloatToOrdered(c.tok_extra[tbn + i]); nc[i] = c.tok_cost[tbn + i]; np[i] = gpn[i];
        }
      }
#pragma unroll
      for (int k = 0; k < FIN_TR; k++) {
        const int i = tid + k * NT;
        if (i < nt) {
          lc[i] = rc[k];
          float b = INFINITY;
          if (f == F) b = MID ? 0.0f : rc[k] + (finals_empty ? 0.0f : d.g.final[rs[k]]) - final_best;   // :430 (MID: :289 extra_cost = 0)
          lx[i] = FloatToOrdered(b);
        }
      }
      // a frame of more than FIN_TR * NT tokens: the rest is read where it lies (an exposed round trip, on the few
      // large frames; holding FIN_CAP / NT tokens per thread in registers spilled the whole pipeline to scratch in the
      // fused kernel -- the records 'prefetched' for the next frame were waited for at once, to be stored).  Their
      // states wait in lp[], which is free until the staging below fills it.
      for (int i = FIN_TR * NT + tid; i < nt; i += NT) {
        const float co = c.tok_cost[tb + i];
        const int st = c.tok_state[tb + i];
        lc[i] = co; lp[i] = st;
        float b = INFINITY;
        if (f == F) b = MID ? 0.0f : co + (finals_empty ? 0.0f : d.g.final[st]) - final_best;
        lx[i] = FloatToOrdered(b);
      }
      LdsBarrier();
      Stamp(&sh, PH_FIN_FETCH);
      // link_extra_cost (:346-350 / :437-441) of a link held as L; x of the last frame reads
      // as +inf above lattice_beam (:462-463)
      auto xval = [&](const u32 *arr, int i, bool last) {
        const float v = OrderedToFloat(arr[i]);
        return (last && v > lattice_beam) ? INFINITY : v;
      };
      auto relax_emit = [&](const LinkLite &L) {
        const int ls = L.sd & 0xFFFFu, ld = L.sd >> 16;
        if (ld == 0xFFFF) return;                    // dropped by the exact cutoff
        float lec = xval(nx, ld, false) + ((lc[ls] + L.ac + L.graph) - nc[ld]);
        if (lec > lattice_beam) return;              // excised (:352)
        if (lec < 0.0f) lec = 0.0f;                  // :360-364
        atomicMin(&lx[ls], FloatToOrdered(lec));
      };
      const int n_reg = min(le - eb, FIN_PF * NT);   // links of this frame held in registers
#pragma unroll
      for (int k = 0; k < FIN_PF; k++) {
        const int li = eb + tid + k * NT;
        if (li >= ee && li < le) relax_emit(rl[k]);
      }
      for (int li = max(ee, eb + n_reg) + tid; li < le; li += NT) relax_emit(Lite(c.links[li], tb, tbn));
      LdsBarrier();
      Stamp(&sh, PH_FIN_EMIT);
      if (ee > eb) {
        const bool last = f == F && !MID;
        auto relax_eps = [&](const LinkLite &L) -> int {
          const int ls = L.sd & 0xFFFFu, ld = L.sd >> 16;
          float lec = xval(lx, ld, last) + ((lc[ls] + L.ac + L.graph) - lc[ld]);
          if (lec > lattice_beam) return 0;
          if (lec < 0.0f) lec = 0.0f;
          const u32 key = FloatToOrdered(lec);
          return atomicMin(&lx[ls], key) > key;
        };
        for (int iter = 0; iter < 20000; iter++) {
          int changed = 0;
#pragma unroll
          for (int k = 0; k < FIN_PF; k++) {
            const int li = eb + tid + k * NT;
            if (li < ee) changed |= relax_eps(rl[k]);
          }
          for (int li = eb + n_reg + tid; li < ee; li += NT) changed |= relax_eps(Lite(c.links[li], tb, tb));
          // workgroup OR through three rotating LDS flags (LDS-only barrier; the flag reset of
          // round k+1 is two barriers away from its last readers)
          if (__any(changed) && (tid & 63) == 0) atomicOr(&s_chg[jit % 3], 1);
          if (tid == 0) s_chg[(jit + 1) % 3] = 0;
          LdsBarrier();
          const int any_changed = s_chg[jit % 3];
          jit++;                      // keeps rotating across frames: a flag left set is reset before its next use
          if (!any_changed) break;
        }
      }
      Stamp(&sh, PH_FIN_EPS);
      if (f == F && !MID) {   // store the clamped values: later reads need no special case
        for (int i = tid; i < nt; i += NT) if (OrderedToFloat(lx[i]) > lattice_beam) lx[i] = INF_O;
        LdsBarrier();
      }
      // ---- PruneTokensForFrame (:492-511) + staging of the survivors.  Staged tokens land in
      // [tok_top - survivors, tok_top): frame f's own records are in registers already and
      // everything above them is dead, so no ordering is needed.
#pragma unroll
      for (int k = 0; k < FIN_TR; k++) {
        const int i = tid + k * NT;
        if (i < nt) {
          const bool alive = lx[i] != INF_O;
          int pos = -1;
          if (alive) { pos = WaveAllocDown(&s_tok_top); c.tok_state[pos] = rs[k]; stage_cost[pos] = rc[k]; }
          lp[i] = pos;
          if (next_hbm_mode) { c.tok_extra[tb + i] = OrderedToFloat(lx[i]); gp[i] = pos; }
        }
      }
      for (int i = FIN_TR * NT + tid; i < nt; i += NT) {      // the part of a large frame that is not in registers
        const bool alive = lx[i] != INF_O;
        int pos = -1;
        if (alive) { pos = WaveAllocDown(&s_tok_top); c.tok_state[pos] = lp[i]; stage_cost[pos] = lc[i]; }
        lp[i] = pos;
        if (next_hbm_mode) { c.tok_extra[tb + i] = OrderedToFloat(lx[i]); gp[i] = pos; }
      }
      LdsBarrier();
      if (tid == 0) new_off[f] = s_tok_top;
      // ---- surviving links (the same expressions, now on final values), remapped.  A staged
      // link lands in [lnk_top - survivors, lnk_top); when that cannot reach this frame's own
      // range (the normal case: the arena has a frame of slack) reads and writes need no
      // ordering, otherwise every chunk is read -> barrier -> written, top-down.
      const bool slack_ok = lnk_top0 - (le - eb) >= le;
      auto survives = [&](const LinkLite &L, bool is_eps) -> bool {
        const int ls = L.sd & 0xFFFFu, ld = L.sd >> 16;
        if (ld == 0xFFFF) return false;
        float lec;
        if (is_eps) lec = OrderedToFloat(lx[ld]) + ((lc[ls] + L.ac + L.graph) - lc[ld]);
        else lec = OrderedToFloat(nx[ld]) + ((lc[ls] + L.ac + L.graph) - nc[ld]);
        return !(lec > lattice_beam);
      };
      auto stage_link = [&](Link L, bool is_eps) {
        const int ps2 = lp[L.src - tb], pd = is_eps ? lp[L.dst - tb] : np[L.dst - tbn];
        if (ps2 < 0 || pd < 0) { atomicOr(&sh.err, ERR_INTERNAL); return; }
        L.src = ps2; L.dst = pd;
        if (!is_eps && !MID) L.ac = L.ac - emit_off;
        c.links[WaveAllocDown(&s_lnk_top)] = L;
      };
      // MID: two passes, the emitting links f -> f+1 first (they end up above the frame's epsilon links: ascending, the
      // arena then reads eps(0) | emit(1) eps(1) | emit(2) ..., the order lnk_off describes), boundaries recorded
      for (int pass = 0; pass < (MID ? 2 : 1); pass++) {
        auto mine = [&](int li) { return !MID || (pass == 0 ? li >= ee : li < ee); };
        if (slack_ok) {
          for (int li = eb + n_reg + tid; li < le; li += NT) {
            const Link L = c.links[li];
            if (mine(li) && survives(Lite(L, tb, li < ee ? tb : tbn), li < ee)) stage_link(L, li < ee);
          }
#pragma unroll
          for (int k = 0; k < FIN_PF; k++) {
            const int li = eb + tid + k * NT;
            if (li < le && mine(li) && survives(rl[k], li < ee)) stage_link(c.links[li], li < ee);
          }
        } else {
          for (int hi = le; hi > eb; hi -= NT) {
            const int li = hi - NT + tid;
            Link L; L.src = -1; L.dst = -1; L.ilabel = 0; L.olabel = 0; L.graph = 0.f; L.ac = 0.f;
            const bool in = li >= eb && li < le && mine(li);
            if (in) L = c.links[li];
            __syncthreads();   // full barrier: global reads before global writes
            if (in && survives(Lite(L, tb, li < ee ? tb : tbn), li < ee)) stage_link(L, li < ee);
          }
        }
        if (MID) {
          __syncthreads();
          const int top = s_lnk_top;       // read by everybody before the next pass allocates again
          __syncthreads();
          if (tid == 0) nlo[pass == 0 ? 2 * f + 2 : 2 * f + 1] = top;
        }
      }
      if (next_hbm_mode) __syncthreads(); else LdsBarrier();   // HBM mode reads tok_extra / positions from global
      Stamp(&sh, PH_FIN_STAGE);
      prev_lds = true;
      cur ^= 1;
    } else {
      // ================= HBM mode (a frame larger than FIN_CAP tokens is involved) =========
      float *xcur = c.tok_extra + tb;
      // (if frame f+1 went through LDS mode it has written its extra costs and positions
      // to HBM: next_hbm_mode was set there)
      for (int i = tid; i < nt; i += NT) {
        float b = INFINITY;
        if (f == F) {
          const float fc = finals_empty ? 0.0f : d.g.final[c.tok_state[tb + i]];
          b = MID ? 0.0f : c.tok_cost[tb + i] + fc - final_best;
        }
        __hip_atomic_store(&bo[i], FloatToOrdered(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      DrainStores();
      __syncthreads();
      for (int li = ee + tid; li < le; li += NT) {
        const Link L = c.links[li];
        if (L.dst < 0) { c.links[li].src = -1; continue; }
        float lec = c.tok_extra[L.dst] + ((c.tok_cost[L.src] + L.ac + L.graph) - c.tok_cost[L.dst]);
        if (lec > lattice_beam) { c.links[li].src = -1; continue; }
        if (lec < 0.0f) lec = 0.0f;
        atomicMin(&bo[L.src - tb], FloatToOrdered(lec));
      }
      __syncthreads();
      for (int i = tid; i < nt; i += NT) {
        const u32 b = LoadU32(&bo[i]);
        float v = OrderedToFloat(b);
        if (f == F && !MID && v > lattice_beam) v = INFINITY;
        xcur[i] = v;
        if (ee > eb) __hip_atomic_store(&xo[i], b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      DrainStores();
      __syncthreads();
      if (ee > eb) {
        for (int iter = 0; iter < 20000; iter++) {
          for (int li = eb + tid; li < ee; li += NT) {
            const Link L = c.links[li];
            float lec = xcur[L.dst - tb] + ((c.tok_cost[L.src] + L.ac + L.graph) - c.tok_cost[L.dst]);
            if (lec > lattice_beam) continue;
            if (lec < 0.0f) lec = 0.0f;
            atomicMin(&xo[L.src - tb], FloatToOrdered(lec));
          }
          __syncthreads();
          int changed = 0, dummy = 0;
          for (int i = tid; i < nt; i += NT) {
            float v = OrderedToFloat(LoadU32(&xo[i]));
            if (f == F && !MID && v > lattice_beam) v = INFINITY;
            if (!(v == xcur[i])) changed = 1;
            xcur[i] = v;
            __hip_atomic_store(&xo[i], LoadU32(&bo[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
          }
          DrainStores();
          BlockSum2(changed, dummy, &sh);
          if (changed == 0) break;
        }
        for (int li = eb + tid; li < ee; li += NT) {
          const Link L = c.links[li];
          const float lec = xcur[L.dst - tb] + ((c.tok_cost[L.src] + L.ac + L.graph) - c.tok_cost[L.dst]);
          if (lec > lattice_beam) c.links[li].src = -1;
        }
      }
      __syncthreads();
      // ---- staging, top-down in chunks (read -> barrier -> write)
      for (int hi = nt; hi > 0; hi -= NT) {
        const int i = hi - NT + tid;
        bool alive = false; int st = 0; float co = 0.f;
        if (i >= 0) { alive = xcur[i] != INFINITY; st = c.tok_state[tb + i]; co = c.tok_cost[tb + i]; }
        __syncthreads();
        if (i >= 0) {
          int pos = -1;
          if (alive) { pos = WaveAllocDown(&s_tok_top); c.tok_state[pos] = st; stage_cost[pos] = co; }
          gp[i] = pos;
        }
      }
      __syncthreads();
      if (tid == 0) new_off[f] = s_tok_top;
      for (int pass = 0; pass < (MID ? 2 : 1); pass++) {
        for (int hi = le; hi > eb; hi -= NT) {
          const int li = hi - NT + tid;
          Link L; L.src = -1; L.dst = -1; L.ilabel = 0; L.olabel = 0; L.graph = 0.f; L.ac = 0.f;
          const bool in = li >= eb && li < le && (!MID || (pass == 0 ? li >= ee : li < ee));
          if (in) L = c.links[li];
          __syncthreads();
          if (in && L.src >= 0 && L.dst >= 0) {
            const bool is_eps = li < ee;
            const int ps2 = gp[L.src - tb];
            const int pd = is_eps ? gp[L.dst - tb] : gpn[L.dst - tbn];
            if (ps2 < 0 || pd < 0) { atomicOr(&sh.err, ERR_INTERNAL); }
            else {
              L.src = ps2; L.dst = pd;
              if (!is_eps && !MID) L.ac = L.ac - emit_off;
              c.links[WaveAllocDown(&s_lnk_top)] = L;
            }
          }
        }
        __syncthreads();
        if (MID && tid == 0) nlo[pass == 0 ? 2 * f + 2 : 2 * f + 1] = s_lnk_top;
      }
      __syncthreads();
      prev_lds = false;
    }
    promote(f - 1);
  }
  __syncthreads();
  Stamp(&sh, PH_FIN_SWEEP);
  const int tok_base = s_tok_top, lnk_base = s_lnk_top;
  const int n_out_tok = c.tok_cap - tok_base, n_out_link = c.lnk_cap - lnk_base;
  __syncthreads();
  for (int f = tid; f <= F + 1; f += NT) c.tok_off[f] = new_off[f] - tok_base;
  __syncthreads();
  if (MID) {
    // ---- move the survivors down to the bottom of the arenas (chunks: read, barrier, write -- a chunk's destination
    // always lies below the source of every later chunk), costs back into tok_cost, link endpoints and offsets rebased
    for (int j = tid; j <= 2 * F + 2; j += NT) c.lnk_off[j] = (j == 0 ? nlo[1] : nlo[j]) - lnk_base;
    for (int b0 = 0; b0 < n_out_tok; b0 += NT) {
      const int i = b0 + tid;
      int st = 0; float co = 0.f;
      if (i < n_out_tok) { st = c.tok_state[tok_base + i]; co = stage_cost[tok_base + i]; }
      __syncthreads();
      if (i < n_out_tok) { c.tok_state[i] = st; c.tok_cost[i] = co; }
    }
    for (int b0 = 0; b0 < n_out_link; b0 += NT) {
      const int i = b0 + tid;
      Link L; L.src = L.dst = L.ilabel = L.olabel = 0; L.graph = L.ac = 0.f;
      if (i < n_out_link) L = c.links[lnk_base + i];
      __syncthreads();
      if (i < n_out_link) { L.src -= tok_base; L.dst -= tok_base; c.links[i] = L; }
    }
    __syncthreads();
    if (tid == 0) {
      S->tok_used = n_out_tok; S->lnk_used = n_out_link;
      S->error |= sh.err;
    }
    return;
  }
  Stamp(&sh, PH_FIN_COMPACT);
  if (tid == 0) {
    for (int i = 0; i < 16; i++) S->phase_cycles[i] += sh.ph[i];
    S->finalized = 1;
    S->final_best_cost = final_best;
    S->final_relative_cost = (best_cost == INFINITY && best_with_final == INFINITY) ? INFINITY : best_with_final - best_cost;
    S->out_ntok = n_out_tok; S->out_nlink = n_out_link;
    S->out_tok_base = tok_base; S->out_lnk_base = lnk_base; S->out_cost_in_map = 1;
    S->error |= sh.err;
  }
}
__global__ __launch_bounds__(NT, 4) void FinalizeKernel2(DecDev d, const int *lanes) {
  __shared__ Sh sh;
  __shared__ FinSh fs;
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_lds[];
  const Ctx c = MakeCtx(d, lanes[blockIdx.x]);
  FinalizeLane2<false, FIN_CAP / NT, 4>(d, c, &sh, &fs, fin_lds);
}
// PruneActiveTokens on un-finalized lanes (kamd_decoder_compact)
__global__ __launch_bounds__(NT, 4) void CompactKernel(DecDev d, const int *lanes) {
  __shared__ Sh sh;
  __shared__ FinSh fs;
  extern __shared__ __attribute__((aligned(16))) unsigned char fin_lds[];
  const Ctx c = MakeCtx(d, lanes[blockIdx.x]);
  FinalizeLane2<true, FIN_CAP / NT, 4>(d, c, &sh, &fs, fin_lds);
}

// ---------------------------------------------------------------- work queue
// Test-set decoding: R resident lanes (one workgroup = one CU each) pull utterances from a
// device-side queue until it is empty, so a launch no longer lasts as long as its longest
// utterance and every CU stays busy while work is left.  This is the GPU form of
// NnetBatchDecoder's decoder threads (nnet3/nnet-batch-compute.cc:1156-1215 Decode(): each
// thread takes the next utterance, InitDecoding / AdvanceDecoding / FinalizeDecoding, hands
// the lattice on) and of decode.sh's --nj jobs (steps/nnet3/decode.sh:96,123).
// Per utterance the lane runs InitLane, AdvanceLane, FinalizeLane2 and then copies the pruned
// raw lattice out of its (reused) arenas into one contiguous blob of a device pool:
//   [frame_off int32 x (F+2)] [tok_state int32 x nt] [tok_cost f32 x nt]
//   [last_final f32 x n_last] [links 24 B x nl]     (link endpoints = lattice-local indices)
// and publishes a record in host-visible memory (system-scope release), which the host polls
// while the kernel is still running: D2H of finished lattices and the host tail (best path,
// determinization) overlap with the search.
struct QueueDev {
  const kamd_queue_task *tasks; int n_tasks;
  int *head;                         // next task to hand out (device)
  int *done_count;                   // finished utterances (device)
  unsigned char *pool; unsigned long long pool_cap; unsigned long long *pool_used;
  kamd_queue_result *results;        // [n_tasks] host-visible, indexed by task.utt
  int *done_ring;                    // [n_tasks] host-visible: utt + 1 in completion order (0 = not yet)
};
enum { ERR_POOL = 64 };

// The three stages are inlined into one body; each of them, and each phase of a frame inside AdvanceLane, takes its own
// view of the two kernel arguments (LoadDecDev / LoadSecondArg), so that nothing but the lane number, the task and the
// utterance id is live from one stage to the next.
KAMD_SEARCH_KERNEL void DecodeQueueKernel(DecDev d_unused, QueueDev q_unused) {
  __shared__ Sh sh;
  __shared__ FinSh fs;
  __shared__ int s_task;
  __shared__ unsigned long long s_off;
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
  const int lane = blockIdx.x;
  for (;;) {
    kamd_decode_task task;
    int utt;
    {
      const QueueDev q = LoadSecondArg<QueueDev>();
      if (Tid() == 0) s_task = __hip_atomic_fetch_add(q.head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int ti = s_task;
      if (ti >= q.n_tasks) break;
      const kamd_queue_task qt = q.tasks[ti];
      task.lane = lane; task.n_frames = qt.n_frames; task.d_loglikes = qt.d_loglikes; task.ld = qt.ld; task.reserved = 0;
      utt = qt.utt;
    }
    {
      const DecDev d = LoadDecDev();
      const Ctx c = MakeCtx(d, Opaque(lane));
      InitLane(d, c, &sh);
    }
    __syncthreads();
    AdvanceLane<true>(lane, &sh, dyn_lds, task);
    __syncthreads();
    {
      const DecDev d = LoadDecDev();
      const Ctx c = MakeCtx(d, Opaque(lane));
      FinalizeLane2<false, 2, 2>(d, c, &sh, &fs, dyn_lds);
    }
    __syncthreads();
    // ---- hand the lattice out
    const int tid = Tid();
    const DecDev d = LoadDecDev();
    const Ctx c = MakeCtx(d, Opaque(lane));
    const QueueDev q = LoadSecondArg<QueueDev>();
    LaneState *S = c.st;
    const int err = S->error, F = S->frame;
    const int nt = err ? 0 : S->out_ntok, nl = err ? 0 : S->out_nlink;
    const int tbase = S->out_tok_base, lbase = S->out_lnk_base;
    const int n_last = err ? 0 : c.tok_off[F + 1] - c.tok_off[F];
    const unsigned long long bytes = err ? 0ull : ((static_cast<unsigned long long>(F + 2) + 2ull * nt + n_last + 6ull * nl) * 4ull + 15ull) & ~15ull;
    if (tid == 0) {
      // bump allocation that takes nothing when the blob does not fit: an utterance too large for what is left fails alone
      // (flag 64) and the smaller ones behind it still find room
      unsigned long long off = 0;
      if (bytes) {
        unsigned long long cur = __hip_atomic_load(q.pool_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (;;) {
          if (cur + bytes > q.pool_cap) { off = q.pool_cap; break; }      // reads as "does not fit" below
          if (__hip_atomic_compare_exchange_strong(q.pool_used, &cur, cur + bytes, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { off = cur; break; }
        }
      }
      s_off = off;
    }
    __syncthreads();
    const unsigned long long off = s_off;
    const bool fits = off + bytes <= q.pool_cap;
    if (bytes && fits) {
      int *o_off = reinterpret_cast<int *>(q.pool + off);
      int *o_state = o_off + (F + 2);
      float *o_cost = reinterpret_cast<float *>(o_state + nt);
      float *o_final = o_cost + nt;
      int *o_link = reinterpret_cast<int *>(o_final + n_last);
      for (int f = tid; f <= F + 1; f += NT) o_off[f] = c.tok_off[f];
      const float *stage_cost = reinterpret_cast<const float *>(c.tok_map);
      for (int i = tid; i < nt; i += NT) { o_state[i] = c.tok_state[tbase + i]; o_cost[i] = stage_cost[tbase + i]; }
      const int lb = c.tok_off[F];
      for (int i = tid; i < n_last; i += NT) o_final[i] = d.g.final[c.tok_state[tbase + lb + i]];
      const int *lsrc = reinterpret_cast<const int *>(c.links + lbase);
      for (int i = tid; i < 6 * nl; i += NT) {
        int v = lsrc[i];
        const int fld = i % 6;
        if (fld < 2) v -= tbase;            // src / dst: arena position -> lattice-local index
        o_link[i] = v;
      }
    }
    // every wavefront's stores must have left the CU before the record is published
    DrainStores();
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // system scope: the blob's stores have reached host memory
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      kamd_queue_result *r = q.results + utt;
      r->error = err | ((bytes && !fits) ? ERR_POOL : 0);
      r->lane = lane; r->n_frames = F; r->n_tok = fits ? nt : 0; r->n_link = fits ? nl : 0; r->n_last = fits ? n_last : 0; r->n_preselected = S->presel_frames;
      r->final_relative_cost = S->final_relative_cost; r->final_best_cost = S->final_best_cost;
      r->blob_off = static_cast<long long>(off); r->blob_bytes = fits ? static_cast<long long>(bytes) : 0;
      for (int i = 0; i < 8; i++) r->counters[i] = S->counters[i];
      for (int i = 0; i < 16; i++) r->phase_cycles[i] = S->phase_cycles[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int k = __hip_atomic_fetch_add(q.done_count, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&r->status, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(&q.done_ring[k], utt + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (err) {
      // an overflow can leave level-2 table words behind that no slot list names: wipe the lane's table
      for (int i = tid; i < d.hash_cap; i += NT) __hip_atomic_store(&c.H[i], EMPTY64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      DrainStores();
    }
    __syncthreads();
  }
}

// Best path WITHOUT finalizing (LatticeFasterOnlineDecoderTpl::BestPathEnd +
// TraceBackBestPath, decoder/lattice-faster-online-decoder.cc:78-165): pick the best
// token of the newest frame (with final-probs if any token is final and use_final_probs),
// then walk back.  No backpointers are stored: the link that produced a token's cost is
// recognised by exact equality  tok_cost[src] + ac + graph == tok_cost[dst]  (the very
// expression that computed it); ties -> smallest link index.
// Output: path arcs in REVERSE order in out_arcs (ilabel, olabel, graph, acoustic-offset).
struct PathArc { int ilabel, olabel; float graph, ac; };
// The walk itself, shared by the best-path and the endpointing kernels.  visit(k, link, emitting, f) is called
// by every thread (uniformly) for the k-th arc from the end; returning false stops the walk.  Returns the number
// of arcs visited, or -1 when no token is alive on the newest frame.
template <typename Visit>
__device__ __forceinline__ int WalkBestPath(const DecDev &d, const Ctx &c, Sh *sh, int use_final_probs, float *final_cost, Visit visit) {
  const int tid = Tid();
  const int F = c.st->frame;
  bool use_final = false;
  if (use_final_probs) {
    float best_cost, best_with_final;
    FinalCosts(d, c, sh, F, &best_cost, &best_with_final);
    use_final = best_with_final != INFINITY;
  }
  u64 key = EMPTY64;
  for (int t = c.tok_off[F] + tid; t < c.tok_off[F + 1]; t += NT) {
    const float fc = use_final ? d.g.final[c.tok_state[t]] : 0.0f;
    const float v = c.tok_cost[t] + fc;
    if (v != INFINITY) {
      const u64 k = (static_cast<u64>(FloatToOrdered(v)) << 32) | static_cast<u32>(t);
      key = k < key ? k : key;
    }
  }
  key = BlockMin64(key, sh);
  if (key == EMPTY64) return -1;
  int cur = static_cast<int>(key & 0xFFFFFFFFu);
  *final_cost = use_final ? d.g.final[c.tok_state[cur]] : 0.0f;
  int n_out = 0, f = F;
  for (int guard = 0; guard < 4 * (F + 2) + 1024; guard++) {
    // the links that can end in 'cur': emitting links of step f in [mb, me), epsilon links of step f in [me, ee)
    const int mb = c.lnk_off[2 * f], me = c.lnk_off[2 * f + 1], ee = c.lnk_off[2 * f + 2];
    const float ccost = c.tok_cost[cur];
    u32 found = 0xFFFFFFFFu;
    for (int li = mb + tid; li < ee; li += NT) {
      const Link L = c.links[li];
      if (L.dst == cur && L.src >= 0 && c.tok_cost[L.src] + L.ac + L.graph == ccost) found = min(found, static_cast<u32>(li));
    }
    const u64 fk = BlockMin64(static_cast<u64>(found), sh);
    if (fk >= 0xFFFFFFFFull) break;              // the start token: nothing produced it
    const Link L = c.links[static_cast<int>(fk)];
    const bool emitting = static_cast<int>(fk) < me;
    const bool go_on = visit(n_out, L, emitting, f);
    n_out++;
    if (!go_on) break;
    cur = L.src;
    if (emitting) f--;
    if (f < 0) break;
  }
  return n_out;
}

__global__ __launch_bounds__(NT, 4) void TracebackKernel(DecDev d, int lane, int use_final_probs,
                                                      PathArc *out_arcs, int out_cap, int *out_n,
                                                      float *out_final_cost) {
  __shared__ Sh sh;
  const Ctx c = MakeCtx(d, lane);
  const int tid = Tid();
  InitSh(&sh);
  float fc = 0.0f;
  const int n = WalkBestPath(d, c, &sh, use_final_probs, &fc, [&](int k, const Link &L, bool emitting, int f) {
    if (tid == 0 && k < out_cap) {
      PathArc a; a.ilabel = L.ilabel; a.olabel = L.olabel; a.graph = L.graph;
      a.ac = emitting ? L.ac - c.cost_offsets[f - 1] : L.ac;
      out_arcs[k] = a;
    }
    return true;
  });
  if (tid == 0) { *out_n = n; if (n >= 0) *out_final_cost = fc; }
}

// The same for a batch of lanes in one launch (a server's partial results after a tick): lane lanes[b] writes its
// arcs at out_arcs + b * out_cap, its count / final cost at head[2b], head[2b+1] (count -1: no token alive).
__global__ __launch_bounds__(NT, 4) void TracebackBatchKernel(DecDev d, const int *lanes, int use_final_probs, PathArc *out_arcs,
                                                           int out_cap, int *head) {
  __shared__ Sh sh;
  const int b = blockIdx.x, tid = Tid();
  const Ctx c = MakeCtx(d, lanes[b]);
  InitSh(&sh);
  PathArc *out = out_arcs + static_cast<size_t>(b) * out_cap;
  float fc = 0.0f;
  int n = -1;
  if (!c.st->error && !c.st->finalized)
    n = WalkBestPath(d, c, &sh, use_final_probs, &fc, [&](int k, const Link &L, bool emitting, int f) {
      if (tid == 0 && k < out_cap) {
        PathArc a; a.ilabel = L.ilabel; a.olabel = L.olabel; a.graph = L.graph;
        a.ac = emitting ? L.ac - c.cost_offsets[f - 1] : L.ac;
        out[k] = a;
      }
      return true;
    });
  if (tid == 0) { head[2 * b] = n; head[2 * b + 1] = __float_as_int(fc); }
}

// TracebackBatchKernel for a server that asks for partial results after EVERY tick: the walk costs a block-wide link
// search per arc (two barriers and a dependent round trip: 0.6-1.2 ms for a 12 s utterance, growing with it), and from
// one tick to the next only the newest few dozen frames of the best path change.  Per lane the previous answer is kept
// on the device -- its arcs OLDEST first in cache[lane][0 .. cache_cap / 2), and for every frame the token (named by its
// HCLG state, as in FrameTraceKernel) that the emitting arc consuming the frame LEFT, with the number of arcs older than
// that arc -- and the walk stops at the first frame whose token is the recorded one: from that token back the path is what
// it was (a token's best incoming link never changes once its frame is closed; pruning keeps the best path's links).
// The new arcs are collected newest first in the upper half of the cache row, moved behind the unchanged prefix, and the
// whole path is copied to out[b] for the host.  head[3b] = arcs on the path (-1: no token alive, -2: the row is too
// small), head[3b + 1] = final cost bits (0: no final-probs), head[3b + 2] = frames now recorded.
__global__ __launch_bounds__(NT, 4) void TracebackIncKernel(DecDev d, const int *lanes, const int *known, PathArc *cache, int cache_cap,
                                                         int *rec, int stride, PathArc *out_arcs, int out_cap, int *head) {
  __shared__ Sh sh;
  const int b = blockIdx.x, tid = Tid(), lane_id = lanes[b];
  const Ctx c = MakeCtx(d, lane_id);
  InitSh(&sh);
  PathArc *P = cache + static_cast<size_t>(lane_id) * cache_cap;
  int *tok = rec + static_cast<size_t>(lane_id) * 4 * stride, *idx = tok + stride, *tstate = idx + stride, *tk = tstate + stride;
  PathArc *out = out_arcs + static_cast<size_t>(b) * out_cap;
  const int n_known = known[b], half = cache_cap / 2;
  float fc = 0.0f;
  int n = -1, matched = -1, newest = -1, oldest = 0;
  if (!c.st->error && !c.st->finalized)
    n = WalkBestPath(d, c, &sh, 0, &fc, [&](int k, const Link &L, bool emitting, int f) {
      if (tid == 0 && k < half) {
        PathArc a; a.ilabel = L.ilabel; a.olabel = L.olabel; a.graph = L.graph;
        a.ac = emitting ? L.ac - c.cost_offsets[f - 1] : L.ac;
        P[cache_cap - 1 - k] = a;
      }
      if (!emitting) return true;
      const int frame = f - 1, state = c.tok_state[L.src];
      if (frame >= stride) return true;
      if (tid == 0) { tstate[frame] = state; tk[frame] = k; }
      if (newest < 0) newest = frame;
      oldest = frame;
      if (frame < n_known && tok[frame] == state) { matched = frame; return false; }
      return true;
    });
  __syncthreads();
  if (n < 0) { if (tid == 0) { head[3 * b] = -1; head[3 * b + 1] = 0; head[3 * b + 2] = 0; } return; }
  const int prefix = matched >= 0 ? idx[matched] : 0;
  const int total = prefix + n;
  if (n > half || total > half || total > out_cap) { if (tid == 0) { head[3 * b] = -2; head[3 * b + 1] = 0; head[3 * b + 2] = 0; } return; }
  for (int k = tid; k < n; k += NT) P[prefix + (n - 1 - k)] = P[cache_cap - 1 - k];        // (lower half <- upper half: disjoint)
  if (newest >= 0)
    for (int fr = oldest + tid; fr <= newest; fr += NT) { tok[fr] = tstate[fr]; idx[fr] = prefix + (n - 1 - tk[fr]); }
  __syncthreads();
  for (int k = tid; k < total; k += NT) out[k] = P[k];
  if (tid == 0) { head[3 * b] = total; head[3 * b + 1] = __float_as_int(fc); head[3 * b + 2] = min(newest + 1, stride); }
}

// What OnlineSilenceWeighting::ComputeCurrentTraceback reads off the decoder (online2/online-ivector-feature.cc:464-510):
// the best path without final-probs, newest frame first, as one (transition-id, token) pair per decoded frame -- the
// emitting arc that consumed the frame and the token it left.  A token is named by its HCLG state: a frame holds one
// token per state and a frame's tokens are only ever deleted, so (frame, state) identifies a token exactly as the
// reference's pointer comparison does.  out[b][k] = {ilabel, state} for frame (count - 1 - k); head[b] = count or -1.
__global__ __launch_bounds__(NT, 4) void FrameTraceKernel(DecDev d, const int *lanes, const int *known, int *prev_tok, int prev_stride,
                                                       int2 *out, int out_cap, int *head) {
  __shared__ Sh sh;
  const int b = blockIdx.x, tid = Tid();
  const Ctx c = MakeCtx(d, lanes[b]);
  InitSh(&sh);
  int2 *o = out + static_cast<size_t>(b) * out_cap;
  // incremental form (prev_tok != NULL): prev[f] = the token recorded for frame f by the previous call, valid for
  // f < known[b].  The walk stops at the first frame whose token is the recorded one -- from there back the path is
  // what it was (the reference's own stopping rule, online-ivector-feature.cc:489-496) -- and that entry is the last
  // one written.  A frame's new token is stored one visit later: by then every thread has passed the barriers of the
  // next link search, i.e. has made its own comparison against the old value.
  int *prev = prev_tok ? prev_tok + static_cast<size_t>(lanes[b]) * prev_stride : NULL;
  const int n_known = prev ? known[b] : 0;
  float fc = 0.0f;
  int n = -1, n_emit = 0, pend_frame = -1, pend_state = 0;
  if (!c.st->error && !c.st->finalized) {
    n = WalkBestPath(d, c, &sh, 0, &fc, [&](int, const Link &L, bool emitting, int f) {
      if (!emitting) return true;
      const int frame = f - 1, state = c.tok_state[L.src];
      if (tid == 0 && pend_frame >= 0) prev[pend_frame] = pend_state;
      pend_frame = -1;
      if (tid == 0 && n_emit < out_cap) o[n_emit] = make_int2(L.ilabel, state);
      n_emit++;
      if (prev) {
        if (frame < n_known && prev[frame] == state) return false;
        if (frame < prev_stride) { pend_frame = frame; pend_state = state; }
      }
      return true;
    });
    if (prev) {
      __syncthreads();
      if (tid == 0 && pend_frame >= 0) prev[pend_frame] = pend_state;
    }
  }
  if (tid == 0) { head[2 * b] = n < 0 ? -1 : c.st->frame; head[2 * b + 1] = n_emit; }
}

// TrailingSilenceLength (online2/online-endpoint.cc:71-102) for a batch of un-finalized lanes: the best path
// without final-probs is walked back from the newest frame, counting transition-ids of silence phones until
// the first one that is not (sil_tid[tid] = 1 for transition-ids of silence phones).
__global__ __launch_bounds__(NT, 4) void TrailingSilenceKernel(DecDev d, const int *lanes, const unsigned char *sil_tid,
                                                            int n_tids, int *out) {
  __shared__ Sh sh;
  const Ctx c = MakeCtx(d, lanes[blockIdx.x]);
  InitSh(&sh);
  int n_sil = 0;
  float fc = 0.0f;
  if (c.st->frame > 0 && !c.st->error) {
    WalkBestPath(d, c, &sh, 0, &fc, [&](int, const Link &L, bool, int) {
      if (L.ilabel == 0) return true;
      if (L.ilabel < n_tids && sil_tid[L.ilabel]) { n_sil++; return true; }
      return false;
    });
  }
  if (Tid() == 0) out[blockIdx.x] = n_sil;
}

static inline size_t AdvanceLdsBytes(int num_pdfs_lds, int lds_table_cap) {
  return 16 + static_cast<size_t>((num_pdfs_lds + 3) & ~3) * 4 + static_cast<size_t>(lds_table_cap) * 8;   // MakeAdvLds
}

// ------------------------------------------------------------------ host
struct Graph {
  GraphDev dev;
  int64_t num_arcs, num_emit;
  int max_ilabel;
  std::vector<void *> allocs;
  std::vector<float> h_final;      // host copy of the final costs (kamd_graph_create): live lattices with final-probs read it
};

struct RawLat {
  std::vector<int32_t> frame, hclg; std::vector<float> cost, fin; std::vector<kamd_lat_arc> arcs;
  int start = -1, frames = 0;
};

struct Decoder {
  DecDev dev;
  Graph *g;
  int num_pdfs = 0;          // 1 + the largest pdf an arc of the graph maps to: every log-likelihood row must be this wide
  kamd_decoder_sizes sizes;
  std::vector<void *> allocs;
  std::vector<LaneState> h_st;
  // HIP-event pairs around the AdvanceKernel launches since the last kamd_decoder_init (a batch may be
  // advanced in several launches: the pipeline overlaps the later nnet slices with the first ones)
  static constexpr int kMaxTimed = 8;
  hipEvent_t ev[2 * kMaxTimed] = {};
  int n_timed = 0;
  float last_ms = 0;
  int *d_lanes = NULL; kamd_decode_task *d_tasks = NULL; int tasks_cap = 0;
  long long *d_tok_base = NULL, *d_lnk_base = NULL; int *d_tok_cap = NULL, *d_lnk_cap = NULL;
  std::vector<long long> h_tok_base, h_lnk_base; std::vector<int> h_tok_cap, h_lnk_cap;
  hipStream_t last_stream = NULL;
  void *d_path = NULL; int path_cap = 0;   // partial best path: {n, final cost, pad} + arcs
  void *d_paths = NULL; size_t paths_cap = 0;   // partial best paths of many lanes (kamd_decoder_partial_best_paths)
  int *d_trace_tok = NULL; int trace_stride = 0; // incremental frame tracebacks: the token recorded per lane and frame ...
  std::vector<int> trace_known;                  // ... and how many frames of it are valid (0 after InitDecoding)
  kamd::PathArc *d_pp_arcs = NULL; int *d_pp_rec = NULL; int pp_cap = 0, pp_stride = 0;   // kamd_decoder_partial_best_paths_incremental: the previous answers
  std::vector<int> pp_known;                     // ... frames recorded per lane (0 after InitDecoding)
  unsigned char *d_sil_tid = NULL; int n_sil_tids = 0; int *d_sil_out = NULL; int sil_out_cap = 0;   // endpointing
  // work queue (kamd_decoder_queue_*)
  unsigned char *d_pool = NULL, *h_pool = NULL; unsigned long long pool_cap = 0;   // the lattice pool: page-locked HOST memory, d_pool = its device address
  unsigned long long *d_pool_used = NULL; int *d_qctl = NULL;       // d_qctl[0] = head, [1] = done count
  kamd::MetaRing qtasks;      // the task list of a work-queue launch (meta_ring.h)
  kamd_queue_result *h_results = NULL; int *h_ring = NULL; int ring_cap = 0;   // host-visible (hipHostMalloc, coherent)
  int q_n = 0, q_next = 0, q_lanes = 0;
  hipEvent_t qev[2] = {};
  hipStream_t q_stream = NULL;
  bool split_uniform = true;
  // host copy of one lane's lattice (canonical), cached by lane
  int cached_lane = -1;
  RawLat live; int live_lane = -1, live_ufp = 1;     // kamd_decoder_live_lattice_size -> kamd_decoder_get_live_raw_lattice
  std::vector<int32_t> lat_frame, lat_hclg; std::vector<float> lat_cost, lat_final;
  std::vector<kamd_lat_arc> lat_arcs; int lat_start = -1, lat_frames = 0;
};

}  // namespace kamd

using kamd::Graph;
using kamd::Decoder;

extern "C" {

kamd_graph *kamd_graph_create(int32_t num_states, int32_t start, const int64_t *arc_off,
                              const kamd_arc *arcs, const float *final_cost) {
  if (num_states <= 0 || start < 0 || start >= num_states) {
    kamd::SetError(KAMD_ERR_ARG, "bad graph (states %d, start %d)", num_states, start);
    return NULL;
  }
  if (!kamd::RequireDevice()) return NULL;
  const int64_t A = arc_off[num_states];
  std::vector<uint2> off(num_states + 1);
  std::vector<kamd_arc> ea, na;
  ea.reserve(A); na.reserve(A / 4 + 1);
  for (int s = 0; s < num_states; s++) {
    off[s].x = static_cast<unsigned>(ea.size()); off[s].y = static_cast<unsigned>(na.size());
    for (int64_t a = arc_off[s]; a < arc_off[s + 1]; a++) {
      if (arcs[a].nextstate < 0 || arcs[a].nextstate >= num_states) {
        kamd::SetError(KAMD_ERR_ARG, "arc %lld: nextstate out of range", static_cast<long long>(a));
        return NULL;
      }
      if (arcs[a].ilabel != 0) ea.push_back(arcs[a]); else na.push_back(arcs[a]);
    }
  }
  off[num_states].x = static_cast<unsigned>(ea.size()); off[num_states].y = static_cast<unsigned>(na.size());
  // (emitting arc indices travel in 31 bits: bit 31 of a candidate's arc word is the target's epsilon flag)
  if (ea.size() > 0x7FFFFFF0u || na.size() > 0xFFFFFFF0u) { kamd::SetError(KAMD_ERR_ARG, "graph too large"); return NULL; }
  // epsilon cycles are illegal (lattice-faster-decoder.cc:997-998): Kahn on the eps graph
  {
    std::vector<int> indeg(num_states, 0);
    for (size_t i = 0; i < na.size(); i++) indeg[na[i].nextstate]++;
    std::vector<int> stack;
    for (int s = 0; s < num_states; s++) if (indeg[s] == 0) stack.push_back(s);
    size_t seen = 0;
    while (!stack.empty()) {
      int s = stack.back(); stack.pop_back(); seen++;
      for (unsigned a = off[s].y; a < off[s + 1].y; a++)
        if (--indeg[na[a].nextstate] == 0) stack.push_back(na[a].nextstate);
    }
    if (seen != static_cast<size_t>(num_states)) {
      kamd::SetError(KAMD_ERR_ARG, "epsilon loops exist in your decoding graph (this is not allowed!)");
      return NULL;
    }
  }
  // device copies: arc targets carry the "has epsilon arcs" flag in bit 31
  for (size_t i = 0; i < ea.size(); i++) { int ns = ea[i].nextstate; if (off[ns + 1].y > off[ns].y) ea[i].nextstate = static_cast<int>(static_cast<unsigned>(ns) | EPS_FLAG); }
  for (size_t i = 0; i < na.size(); i++) { int ns = na[i].nextstate; if (off[ns + 1].y > off[ns].y) na[i].nextstate = static_cast<int>(static_cast<unsigned>(ns) | EPS_FLAG); }
  Graph *g = new Graph();
  g->num_arcs = A; g->num_emit = static_cast<int64_t>(ea.size());
  g->max_ilabel = 0;
  for (size_t i = 0; i < ea.size(); i++) g->max_ilabel = std::max(g->max_ilabel, ea[i].ilabel);
  auto up = [&](const void *src, size_t bytes) -> void * {
    void *p = NULL;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return NULL;
    g->allocs.push_back(p);
    if (bytes && hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return p;
  };
  g->h_final.assign(final_cost, final_cost + num_states);
  g->dev.num_states = num_states; g->dev.start = start;
  g->dev.start_flagged = (off[start + 1].y > off[start].y) ? static_cast<int>(static_cast<unsigned>(start) | EPS_FLAG) : start;
  g->dev.off = static_cast<const uint2 *>(up(off.data(), off.size() * sizeof(uint2)));
  g->dev.e_arcs = static_cast<const kamd_arc *>(up(ea.data(), ea.size() * sizeof(kamd_arc)));
  g->dev.n_arcs = static_cast<const kamd_arc *>(up(na.data(), na.size() * sizeof(kamd_arc)));
  g->dev.final = static_cast<const float *>(up(final_cost, num_states * sizeof(float)));
  if (!g->dev.off || !g->dev.e_arcs || !g->dev.n_arcs || !g->dev.final) {
    kamd::SetError(KAMD_ERR_HIP, "graph upload failed");
    kamd_graph_destroy(reinterpret_cast<kamd_graph *>(g));
    return NULL;
  }
  return reinterpret_cast<kamd_graph *>(g);
}
void kamd_graph_destroy(kamd_graph *h) {
  Graph *g = reinterpret_cast<Graph *>(h);
  if (!g) return;
  for (size_t i = 0; i < g->allocs.size(); i++) (void)hipFree(g->allocs[i]);
  delete g;
}
int32_t kamd_graph_num_states(const kamd_graph *h) { return reinterpret_cast<const Graph *>(h)->dev.num_states; }
int64_t kamd_graph_num_arcs(const kamd_graph *h) { return reinterpret_cast<const Graph *>(h)->num_arcs; }

static int CheckConfig(const kamd_decoder_config *c) {
  // LatticeFasterDecoderConfig::Check (lattice-faster-decoder.h:84-89)
  if (!(c->beam > 0.0f && c->max_active > 1 && c->lattice_beam > 0.0f && c->min_active <= c->max_active &&
        c->prune_interval > 0 && c->beam_delta > 0.0f && c->hash_ratio >= 1.0f && c->prune_scale > 0.0f &&
        c->prune_scale < 1.0f && c->min_active >= 0))
    return kamd::SetError(KAMD_ERR_ARG, "invalid LatticeFasterDecoderConfig");
  return KAMD_OK;
}

kamd_decoder *kamd_decoder_create(const kamd_graph *gh, const kamd_decoder_config *cfg,
                                  const kamd_decoder_sizes *sz, const int32_t *tid2pdf, int32_t num_tids) {
  if (CheckConfig(cfg) != KAMD_OK || !kamd::RequireDevice()) return NULL;
  kamd_decoder_sizes s;
  if (sz) s = *sz; else kamd_decoder_sizes_default(&s);
  if (s.max_lanes < 1 || s.hash_capacity < 64 || (s.hash_capacity & (s.hash_capacity - 1)) ||
      s.arena_tokens < 16 || s.arena_links < 16 || s.max_frames < 1 || s.arena_tokens > 2000000000LL ||
      s.arena_links > 2000000000LL || s.hash_capacity < s.max_frames + 2) {
    kamd::SetError(KAMD_ERR_ARG, "bad decoder sizes");
    return NULL;
  }
  Decoder *D = new Decoder();
  D->g = const_cast<Graph *>(reinterpret_cast<const Graph *>(gh));   // (never written through: one graph serves many decoder objects and threads)
  D->sizes = s;
  kamd::DecDev &d = D->dev;
  memset(&d, 0, sizeof(d));
  d.g = D->g->dev; d.cfg = *cfg;
  d.hash_cap = s.hash_capacity; d.hash_mask = s.hash_capacity - 1; d.max_frames = s.max_frames;
  const size_t L = s.max_lanes, hc = static_cast<size_t>(s.hash_capacity) + LDS_TABLE_CAP, at = s.arena_tokens, al = s.arena_links, mf = s.max_frames;
  bool ok = true;
  auto alloc = [&](size_t bytes, int fill) -> void * {
    void *p = NULL;
    if (hipMalloc(&p, bytes) != hipSuccess) { ok = false; return NULL; }
    D->allocs.push_back(p);
    if (fill >= 0 && hipMemset(p, fill, bytes) != hipSuccess) ok = false;
    return p;
  };
  d.H = static_cast<kamd::u64 *>(alloc(L * static_cast<size_t>(s.hash_capacity) * 8, 0xFF));
  d.e2 = static_cast<kamd::u64 *>(alloc(L * static_cast<size_t>(s.hash_capacity) * 8, -1));
  d.slots = static_cast<kamd::u32 *>(alloc(L * hc * 4, 0));
  d.slot_tok = static_cast<int *>(alloc(L * hc * 4, 0));
  d.stamp = static_cast<kamd::u32 *>(alloc(L * hc * 4, 0));
  d.wl = static_cast<kamd::u32 *>(alloc(L * hc * 8, 0));
  d.scratch = static_cast<float *>(alloc(L * hc * 8, 0));
  d.tok_state = static_cast<int *>(alloc(L * at * 4, -1));
  d.tok_cost = static_cast<float *>(alloc(L * at * 4, -1));
  d.tok_extra = static_cast<float *>(alloc(L * at * 4, -1));
  d.tok_map = static_cast<int *>(alloc(L * at * 4, -1));
  d.links = static_cast<kamd::Link *>(alloc(L * al * sizeof(kamd::Link), -1));
  d.tok_off = static_cast<int *>(alloc(L * (mf + 2) * 4, 0));
  d.lnk_off = static_cast<int *>(alloc(L * (2 * (mf + 2) + 1) * 4, 0));
  d.cost_offsets = static_cast<float *>(alloc(L * (mf + 1) * 4, 0));
  d.trace_ntok = static_cast<int *>(alloc(L * (mf + 1) * 4, 0));
  d.trace_cutoff = static_cast<float *>(alloc(L * (mf + 1) * 4, 0));
  d.st = static_cast<kamd::LaneState *>(alloc(L * sizeof(kamd::LaneState), 0));
  D->d_tok_base = static_cast<long long *>(alloc(L * 8, 0)); D->d_lnk_base = static_cast<long long *>(alloc(L * 8, 0));
  D->d_tok_cap = static_cast<int *>(alloc(L * 4, 0)); D->d_lnk_cap = static_cast<int *>(alloc(L * 4, 0));
  d.lane_tok_base = D->d_tok_base; d.lane_lnk_base = D->d_lnk_base;
  d.lane_tok_cap = D->d_tok_cap; d.lane_lnk_cap = D->d_lnk_cap;
  D->h_tok_base.resize(L); D->h_lnk_base.resize(L); D->h_tok_cap.resize(L); D->h_lnk_cap.resize(L);
  for (size_t l = 0; l < L; l++) {   // default: uniform partition of the pools
    D->h_tok_base[l] = static_cast<long long>(l * at); D->h_lnk_base[l] = static_cast<long long>(l * al);
    D->h_tok_cap[l] = static_cast<int>(at); D->h_lnk_cap[l] = static_cast<int>(al);
  }
  if (ok) {
    if (hipMemcpy(D->d_tok_base, D->h_tok_base.data(), L * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(D->d_lnk_base, D->h_lnk_base.data(), L * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(D->d_tok_cap, D->h_tok_cap.data(), L * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(D->d_lnk_cap, D->h_lnk_cap.data(), L * 4, hipMemcpyHostToDevice) != hipSuccess) ok = false;
  }
  int num_pdfs = D->g->max_ilabel;   // identity map: pdf = ilabel - 1
  if (tid2pdf) {
    if (D->g->max_ilabel > num_tids) {
      kamd::SetError(KAMD_ERR_ARG, "graph ilabel %d exceeds the transition-id table (%d)", D->g->max_ilabel, num_tids);
      kamd_decoder_destroy(reinterpret_cast<kamd_decoder *>(D));
      return NULL;
    }
    int *p = static_cast<int *>(alloc((static_cast<size_t>(num_tids) + 1) * 4, 0));
    if (p && hipMemcpy(p, tid2pdf, (static_cast<size_t>(num_tids) + 1) * 4, hipMemcpyHostToDevice) != hipSuccess) ok = false;
    d.tid2pdf = p;
    num_pdfs = 0;
    for (int t = 1; t <= num_tids; t++) num_pdfs = std::max(num_pdfs, tid2pdf[t] + 1);
  }
  D->num_pdfs = num_pdfs;
  {
    uint2 *ep = static_cast<uint2 *>(alloc(static_cast<size_t>(std::max<int64_t>(D->g->num_emit, 1)) * 8, 0));
    d.e_hot = ep;
    if (ok && D->g->num_emit > 0) {
      hipLaunchKernelGGL(kamd::ArcHotKernel, dim3(kamd::CeilDiv(D->g->num_emit, 256)), dim3(256), 0, 0, d.g.e_arcs,
                         static_cast<long long>(D->g->num_emit), d.tid2pdf, ep);
      if (hipDeviceSynchronize() != hipSuccess) ok = false;
    }
  }
  // LDS budget (160 KB per CU): static Sh + the level-1 table region (whose upper half doubles as the phases' scratch:
  // MakeAdvLds) + as much of the log-likelihood row as still fits (pdfs beyond that are read from HBM: LogLikePdf)
  d.lds_table_cap = LDS_TABLE_CAP;
  d.big_frame_tokens = BIG_FRAME_TOKENS;
  if (const char *e = getenv("KAMD_BIG_FRAME_TOKENS")) d.big_frame_tokens = atoi(e);      // (experiments: tools/ab_bench.py)
  if (const char *e = getenv("KAMD_GOOD_FIRST")) d.good_first = static_cast<float>(atof(e));
  d.preselect = 1;
  if (const char *e = getenv("KAMD_PRESELECT")) d.preselect = atoi(e);
  d.ps_margin_pct = 8;
  d.ps_adapt = 1;
  if (const char *e = getenv("KAMD_PS_ADAPT")) d.ps_adapt = atoi(e) != 0;
  d.ps_worth_pct = 150;
  if (const char *e = getenv("KAMD_PS_WORTH_PCT")) d.ps_worth_pct = std::max(100, std::min(100000, atoi(e)));
  if (const char *e = getenv("KAMD_PS_MARGIN_PCT")) d.ps_margin_pct = std::max(1, std::min(400, atoi(e)));
  d.num_pdfs_lds = 0;
  const size_t lds_budget = 160 * 1024 / LANES_PER_CU - sizeof(kamd::Sh) - 1024;
  const size_t fixed = kamd::AdvanceLdsBytes(0, LDS_TABLE_CAP);
  if (fixed < lds_budget) d.num_pdfs_lds = static_cast<int>(std::min<size_t>(static_cast<size_t>(num_pdfs), (lds_budget - fixed) / 4) & ~static_cast<size_t>(3));
  // the row is filled by LDS-DMA, whose LDS base travels in M0[15:0]: keep its end below 64 KB (the row starts 16 bytes
  // into the dynamic part, which follows the kernels' static LDS: Sh, FinSh and the queue kernel's two words)
  const size_t static_lds = (sizeof(kamd::Sh) + sizeof(kamd::FinSh) + 64 + 255) & ~static_cast<size_t>(255);
  static_assert(sizeof(kamd::Sh) + sizeof(kamd::FinSh) + 64 < 16 * 1024, "static LDS of the search kernels");
  const size_t dma_reach = (65536 - static_lds - 16) / 4;
  if (static_cast<size_t>(d.num_pdfs_lds) > dma_reach) d.num_pdfs_lds = static_cast<int>(dma_reach & ~static_cast<size_t>(3));
  if (d.num_pdfs_lds + 3 >= num_pdfs && static_cast<size_t>(num_pdfs) <= dma_reach && kamd::AdvanceLdsBytes(num_pdfs, LDS_TABLE_CAP) <= lds_budget) d.num_pdfs_lds = num_pdfs;
  if (ok && (hipFuncSetAttribute(reinterpret_cast<const void *>(kamd::FinalizeKernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 6 * FIN_CAP * 4) != hipSuccess ||
             hipFuncSetAttribute(reinterpret_cast<const void *>(kamd::FinalizeKernel2),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LDS_BYTES) != hipSuccess ||
             hipFuncSetAttribute(reinterpret_cast<const void *>(kamd::CompactKernel),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, FIN_LDS_BYTES) != hipSuccess))
    ok = false;
  if (ok && hipFuncSetAttribute(reinterpret_cast<const void *>(kamd::AdvanceKernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(kamd::AdvanceLdsBytes(d.num_pdfs_lds, LDS_TABLE_CAP))) != hipSuccess)
    ok = false;
  if (ok && hipFuncSetAttribute(reinterpret_cast<const void *>(kamd::DecodeQueueKernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(std::max<size_t>(kamd::AdvanceLdsBytes(d.num_pdfs_lds, LDS_TABLE_CAP), FIN_LDS_BYTES))) != hipSuccess)
    ok = false;
  for (int i = 0; ok && i < 2 * Decoder::kMaxTimed; i++) if (hipEventCreate(&D->ev[i]) != hipSuccess) ok = false;
  for (int i = 0; ok && i < 2; i++) if (hipEventCreate(&D->qev[i]) != hipSuccess) ok = false;
  if (!ok) {
    kamd::SetError(KAMD_ERR_HIP, "decoder allocation failed (%zu lanes): %s", L, hipGetErrorString(hipGetLastError()));
    kamd_decoder_destroy(reinterpret_cast<kamd_decoder *>(D));
    return NULL;
  }
  D->h_st.resize(L);
  return reinterpret_cast<kamd_decoder *>(D);
}

void kamd_decoder_destroy(kamd_decoder *h) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (!D) return;
  for (size_t i = 0; i < D->allocs.size(); i++) (void)hipFree(D->allocs[i]);
  if (D->d_path) (void)hipFree(D->d_path);
  if (D->d_trace_tok) (void)hipFree(D->d_trace_tok);
  if (D->d_pp_arcs) (void)hipFree(D->d_pp_arcs);
  if (D->d_pp_rec) (void)hipFree(D->d_pp_rec);
  if (D->d_paths) (void)hipFree(D->d_paths);
  if (D->d_sil_tid) (void)hipFree(D->d_sil_tid);
  if (D->d_sil_out) (void)hipFree(D->d_sil_out);
  if (D->d_lanes) (void)hipFree(D->d_lanes);
  if (D->d_tasks) (void)hipFree(D->d_tasks);
  for (int i = 0; i < 2 * Decoder::kMaxTimed; i++) if (D->ev[i]) (void)hipEventDestroy(D->ev[i]);
  if (D->h_pool) (void)hipHostFree(D->h_pool);
  if (D->d_pool_used) (void)hipFree(D->d_pool_used);
  if (D->d_qctl) (void)hipFree(D->d_qctl);
  if (D->h_results) (void)hipHostFree(D->h_results);
  if (D->h_ring) (void)hipHostFree(D->h_ring);
  for (int i = 0; i < 2; i++) if (D->qev[i]) (void)hipEventDestroy(D->qev[i]);
  delete D;
}

int kamd_decoder_reserve(kamd_decoder *h, const int32_t *lane_frames, int n) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  const size_t L = D->sizes.max_lanes;
  if (n < 0 || static_cast<size_t>(n) > L) return kamd::SetError(KAMD_ERR_ARG, "bad lane count");
  const double pool_t = static_cast<double>(L) * D->sizes.arena_tokens, pool_l = static_cast<double>(L) * D->sizes.arena_links;
  double tot = 0;
  for (int i = 0; i < n; i++) {
    if (lane_frames[i] < 0 || lane_frames[i] > D->sizes.max_frames) return kamd::SetError(KAMD_ERR_ARG, "lane %d: %d frames > max_frames", i, lane_frames[i]);
    tot += lane_frames[i] + 2;
  }
  long long tb = 0, lb = 0;
  for (size_t l = 0; l < L; l++) {
    double share = (l < static_cast<size_t>(n) && tot > 0) ? (lane_frames[l] + 2) / tot : 0.0;
    long long tc = static_cast<long long>(pool_t * share), lc = static_cast<long long>(pool_l * share);
    if (tc > 2000000000LL) tc = 2000000000LL;
    if (lc > 2000000000LL) lc = 2000000000LL;
    D->h_tok_base[l] = tb; D->h_lnk_base[l] = lb;
    D->h_tok_cap[l] = static_cast<int>(tc); D->h_lnk_cap[l] = static_cast<int>(lc);
    tb += tc; lb += lc;
  }
  KAMD_HIP(hipMemcpy(D->d_tok_base, D->h_tok_base.data(), L * 8, hipMemcpyHostToDevice));
  KAMD_HIP(hipMemcpy(D->d_lnk_base, D->h_lnk_base.data(), L * 8, hipMemcpyHostToDevice));
  KAMD_HIP(hipMemcpy(D->d_tok_cap, D->h_tok_cap.data(), L * 4, hipMemcpyHostToDevice));
  KAMD_HIP(hipMemcpy(D->d_lnk_cap, D->h_lnk_cap.data(), L * 4, hipMemcpyHostToDevice));
  D->split_uniform = false;
  return KAMD_OK;
}

static int ReserveUniform(Decoder *D) {
  if (D->split_uniform) return KAMD_OK;
  const size_t L = D->sizes.max_lanes, at = D->sizes.arena_tokens, al = D->sizes.arena_links;
  for (size_t l = 0; l < L; l++) {
    D->h_tok_base[l] = static_cast<long long>(l * at); D->h_lnk_base[l] = static_cast<long long>(l * al);
    D->h_tok_cap[l] = static_cast<int>(at); D->h_lnk_cap[l] = static_cast<int>(al);
  }
  KAMD_HIP(hipMemcpy(D->d_tok_base, D->h_tok_base.data(), L * 8, hipMemcpyHostToDevice));
  KAMD_HIP(hipMemcpy(D->d_lnk_base, D->h_lnk_base.data(), L * 8, hipMemcpyHostToDevice));
  KAMD_HIP(hipMemcpy(D->d_tok_cap, D->h_tok_cap.data(), L * 4, hipMemcpyHostToDevice));
  KAMD_HIP(hipMemcpy(D->d_lnk_cap, D->h_lnk_cap.data(), L * 4, hipMemcpyHostToDevice));
  D->split_uniform = true;
  return KAMD_OK;
}

int kamd_decoder_lanes_per_cu(void) { return LANES_PER_CU; }

int kamd_decoder_lds_layout(const kamd_decoder *h, int32_t *num_pdfs_lds, int32_t *table_words) {
  const Decoder *D = reinterpret_cast<const Decoder *>(h);
  if (num_pdfs_lds) *num_pdfs_lds = D->dev.num_pdfs_lds;
  if (table_words) *table_words = D->dev.lds_table_cap;
  return KAMD_OK;
}

int kamd_decoder_set_level1_table(kamd_decoder *h, int32_t words) {
  if (words < 0 || words > LDS_TABLE_CAP || (words & (words - 1)) != 0 || (words > 0 && words < 64))
    return kamd::SetError(KAMD_ERR_ARG, "level-1 table region: 0 or a power of two in [64, %d] words", LDS_TABLE_CAP);
  Decoder *D = reinterpret_cast<Decoder *>(h);
  D->dev.lds_table_cap = words;       // (the LDS reservation stays what it was: the region is only used less)
  D->dev.big_frame_tokens = words >= LDS_TABLE_CAP ? BIG_FRAME_TOKENS : (words / 2) * 3 / 4;      // three quarters of the half-region table
  return KAMD_OK;
}

int kamd_decoder_set_token_preselection(kamd_decoder *h, int on) {
  reinterpret_cast<Decoder *>(h)->dev.preselect = on != 0;
  return KAMD_OK;
}

int kamd_decoder_set_search_mode(kamd_decoder *h, int mode) {
  if (mode != 1 && mode != 2) return kamd::SetError(KAMD_ERR_ARG, "search mode must be 1 (canonical) or 2 (canonical-loose)");
  reinterpret_cast<Decoder *>(h)->dev.loose = mode == 2;
  return KAMD_OK;
}

int kamd_decoder_set_options(kamd_decoder *h, const kamd_decoder_config *cfg) {
  if (CheckConfig(cfg) != KAMD_OK) return KAMD_ERR_ARG;
  reinterpret_cast<Decoder *>(h)->dev.cfg = *cfg;
  return KAMD_OK;
}

static int EnsureTaskBuf(Decoder *D, int n) {
  if (n <= D->tasks_cap) return KAMD_OK;
  if (D->d_lanes) (void)hipFree(D->d_lanes);
  if (D->d_tasks) (void)hipFree(D->d_tasks);
  D->d_lanes = NULL; D->d_tasks = NULL;
  int cap = std::max(n, 64);
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_lanes), cap * sizeof(int)));
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_tasks), cap * sizeof(kamd_decode_task)));
  D->tasks_cap = cap;
  return KAMD_OK;
}

static int CheckLanes(Decoder *D, const int32_t *lanes, int n) {
  for (int i = 0; i < n; i++)
    if (lanes[i] < 0 || lanes[i] >= D->sizes.max_lanes) return kamd::SetError(KAMD_ERR_ARG, "lane %d out of range", lanes[i]);
  return KAMD_OK;
}

int kamd_decoder_init(kamd_decoder *h, const int32_t *lanes, int n, void *stream) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (CheckLanes(D, lanes, n) != KAMD_OK || EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KAMD_HIP(hipMemcpyAsync(D->d_lanes, lanes, n * sizeof(int), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(kamd::InitKernel, dim3(n), dim3(NT), 0, st, D->dev, D->d_lanes);
  KAMD_HIP(hipGetLastError());
  D->last_stream = st; D->cached_lane = -1; D->n_timed = 0;
  if (!D->trace_known.empty()) for (int i = 0; i < n; i++) D->trace_known[lanes[i]] = 0;     // a new utterance: nothing recorded
  if (!D->pp_known.empty()) for (int i = 0; i < n; i++) D->pp_known[lanes[i]] = 0;
  return KAMD_OK;
}

int kamd_decoder_advance(kamd_decoder *h, const kamd_decode_task *tasks, int n, void *stream) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_HIP;
  for (int i = 0; i < n; i++)
    if (tasks[i].lane < 0 || tasks[i].lane >= D->sizes.max_lanes || tasks[i].n_frames < 0)
      return kamd::SetError(KAMD_ERR_ARG, "task %d: bad lane / frame count", i);
  for (int i = 0; i < n; i++)
    if (tasks[i].n_frames > 0 && tasks[i].ld < D->num_pdfs)
      return kamd::SetError(KAMD_ERR_ARG, "task %d: log-likelihood rows of %d columns, the graph's arcs map to pdfs up to %d", i, tasks[i].ld, D->num_pdfs - 1);
  hipStream_t st = static_cast<hipStream_t>(stream);
  // longest first: the tail of the launch is the longest utterance, start it early
  std::vector<kamd_decode_task> sorted(tasks, tasks + n);
  std::stable_sort(sorted.begin(), sorted.end(),
                   [](const kamd_decode_task &a, const kamd_decode_task &b) { return a.n_frames > b.n_frames; });
  KAMD_HIP(hipMemcpyAsync(D->d_tasks, sorted.data(), n * sizeof(kamd_decode_task), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  const int slot = D->n_timed < Decoder::kMaxTimed ? D->n_timed : Decoder::kMaxTimed - 1;   // streaming: the last pair is reused
  KAMD_HIP(hipEventRecord(D->ev[2 * slot], st));
  const size_t lds = kamd::AdvanceLdsBytes(D->dev.num_pdfs_lds, LDS_TABLE_CAP);
  hipLaunchKernelGGL(kamd::AdvanceKernel, dim3(n), dim3(NT), lds, st, D->dev, D->d_tasks);
  KAMD_HIP(hipGetLastError());
  KAMD_HIP(hipEventRecord(D->ev[2 * slot + 1], st));
  D->n_timed = slot + 1; D->last_stream = st; D->cached_lane = -1;
  return KAMD_OK;
}

int kamd_decoder_finalize(kamd_decoder *h, const int32_t *lanes, int n, void *stream) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (CheckLanes(D, lanes, n) != KAMD_OK || EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KAMD_HIP(hipMemcpyAsync(D->d_lanes, lanes, n * sizeof(int), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  // KAMD_FINALIZE_V1=1 selects the first-generation kernel (mark, then compact in place)
  static const bool v1 = getenv("KAMD_FINALIZE_V1") != NULL && getenv("KAMD_FINALIZE_V1")[0] == '1';
  if (v1) hipLaunchKernelGGL(kamd::FinalizeKernel, dim3(n), dim3(NT), 6 * FIN_CAP * 4, st, D->dev, D->d_lanes);
  else hipLaunchKernelGGL(kamd::FinalizeKernel2, dim3(n), dim3(NT), FIN_LDS_BYTES, st, D->dev, D->d_lanes);
  KAMD_HIP(hipGetLastError());
  D->last_stream = st; D->cached_lane = -1;
  return KAMD_OK;
}

// PruneActiveTokens for un-finalized lanes (lattice-faster-decoder.cc:519-546): what the final sweep would drop anyway
// is dropped now and the survivors move to the bottom of the lane's arenas; decoding goes on.  The final lattice is
// the one an uncompacted decode gives.  A stream calls it when its arena fills up (kamd_decoder_lane_usage).
int kamd_decoder_compact(kamd_decoder *h, const int32_t *lanes, int n, void *stream) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (CheckLanes(D, lanes, n) != KAMD_OK || EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_ARG;
  hipStream_t st = static_cast<hipStream_t>(stream);
  KAMD_HIP(hipMemcpyAsync(D->d_lanes, lanes, n * sizeof(int), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(kamd::CompactKernel, dim3(n), dim3(NT), FIN_LDS_BYTES, st, D->dev, D->d_lanes);
  KAMD_HIP(hipGetLastError());
  D->last_stream = st; D->cached_lane = -1;
  return KAMD_OK;
}

// as of the last kamd_decoder_sync: records in use and capacity of the lane's token and link arenas
int kamd_decoder_lane_usage(kamd_decoder *h, int lane, int32_t *tok_used, int32_t *tok_cap, int32_t *lnk_used, int32_t *lnk_cap) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (CheckLanes(D, &lane, 1) != KAMD_OK) return KAMD_ERR_ARG;
  *tok_used = D->h_st[lane].tok_used; *lnk_used = D->h_st[lane].lnk_used;
  *tok_cap = D->h_tok_cap[lane]; *lnk_cap = D->h_lnk_cap[lane];
  return KAMD_OK;
}

int kamd_decoder_sync(kamd_decoder *h) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  KAMD_HIP(hipStreamSynchronize(D->last_stream));
  if (D->n_timed > 0) {
    float sum = 0;
    for (int i = 0; i < D->n_timed; i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, D->ev[2 * i], D->ev[2 * i + 1]) == hipSuccess) sum += ms;
    }
    D->last_ms = sum;
  }
  KAMD_HIP(hipMemcpy(D->h_st.data(), D->dev.st, D->h_st.size() * sizeof(kamd::LaneState), hipMemcpyDeviceToHost));
  for (size_t l = 0; l < D->h_st.size(); l++) {
    int e = D->h_st[l].error;
    if (e)
      return kamd::SetError(KAMD_ERR_CAPACITY,
                            "lane %zu: device capacity exceeded (flags %d:%s%s%s%s%s%s) at frame %d; raise kamd_decoder_sizes",
                            l, e, (e & 1) ? " hash" : "", (e & 2) ? " token-arena" : "", (e & 4) ? " link-arena" : "",
                            (e & 8) ? " max-frames" : "", (e & 16) ? " worklist" : "", (e & 32) ? " internal" : "",
                            D->h_st[l].frame);
  }
  return KAMD_OK;
}

// kamd_decoder_sync restricted to the lanes of interest: blocks, refreshes the host copy of every lane's state, and
// reports the capacity flags of lanes[i] in lane_error[i] instead of failing the call (a server keeps decoding its
// other streams when one of them overflows).  Returns KAMD_ERR_HIP only for runtime failures.
int kamd_decoder_sync_lanes(kamd_decoder *h, const int32_t *lanes, int n, int32_t *lane_error) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (CheckLanes(D, lanes, n) != KAMD_OK) return KAMD_ERR_ARG;
  KAMD_HIP(hipStreamSynchronize(D->last_stream));
  if (D->n_timed > 0) {
    float sum = 0;
    for (int i = 0; i < D->n_timed; i++) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, D->ev[2 * i], D->ev[2 * i + 1]) == hipSuccess) sum += ms;
    }
    D->last_ms = sum;
  }
  KAMD_HIP(hipMemcpy(D->h_st.data(), D->dev.st, D->h_st.size() * sizeof(kamd::LaneState), hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) if (lane_error) lane_error[i] = D->h_st[lanes[i]].error;
  return KAMD_OK;
}

float kamd_decoder_last_advance_ms(kamd_decoder *h) { return reinterpret_cast<Decoder *>(h)->last_ms; }
int kamd_decoder_last_advance_launches(kamd_decoder *h) { return reinterpret_cast<Decoder *>(h)->n_timed; }

static int LaneOk(Decoder *D, int lane) {
  if (lane < 0 || lane >= D->sizes.max_lanes) return kamd::SetError(KAMD_ERR_ARG, "lane %d out of range", lane);
  return KAMD_OK;
}

int kamd_decoder_num_frames_decoded(kamd_decoder *h, int lane) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (LaneOk(D, lane) != KAMD_OK) return KAMD_ERR_ARG;
  return D->h_st[lane].frame;
}
float kamd_decoder_final_relative_cost(kamd_decoder *h, int lane) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (LaneOk(D, lane) != KAMD_OK) return INFINITY;
  return D->h_st[lane].final_relative_cost;
}
int kamd_decoder_reached_final(kamd_decoder *h, int lane) {
  // ReachedFinal(): FinalRelativeCost() != infinity (lattice-faster-decoder.h:283-285)
  return kamd_decoder_final_relative_cost(h, lane) != INFINITY;
}
int kamd_decoder_get_counters(kamd_decoder *h, int lane, int64_t counters[8]) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (LaneOk(D, lane) != KAMD_OK) return KAMD_ERR_ARG;
  for (int i = 0; i < 8; i++) counters[i] = D->h_st[lane].counters[i];
  return KAMD_OK;
}
int kamd_decoder_partial_best_path(kamd_decoder *h, int lane, int use_final_probs, int32_t *alignment,
                                   int ali_cap, int *ali_len, int32_t *words, int words_cap, int *words_len,
                                   float *graph_cost, float *acoustic_cost) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (LaneOk(D, lane) != KAMD_OK) return KAMD_ERR_ARG;
  if (D->h_st[lane].finalized) return kamd::SetError(KAMD_ERR_STATE, "lane %d is finalized: use kamd_decoder_best_path", lane);
  if (!use_final_probs) {
    // GetBestPath(end_of_utterance = false) after every chunk: the incremental walk (only the frames whose token changed)
    int32_t al = 0, wl = 0;
    const int32_t one = lane;
    const int rc = kamd_decoder_partial_best_paths_incremental(h, &one, 1, alignment, ali_cap, &al, words, words_cap, &wl, graph_cost, acoustic_cost);
    *ali_len = 0; *words_len = 0;
    if (rc != KAMD_OK) return rc;
    if (al < 0) { *graph_cost = INFINITY; *acoustic_cost = INFINITY; return kamd::SetError(KAMD_ERR_STATE, "no tokens alive on the newest frame"); }
    *ali_len = al; *words_len = wl;
    return KAMD_OK;
  }
  const int cap = 4 * (D->h_st[lane].frame + 2) + 1024;
  if (cap > D->path_cap) {         // one buffer for the decoder's lifetime (a server asks for partial results every tick)
    const int grow = std::max(cap, 2 * D->path_cap);
    if (D->d_path) (void)hipFree(D->d_path);
    D->d_path = NULL; D->path_cap = 0;
    KAMD_HIP(hipMalloc(&D->d_path, 16 + static_cast<size_t>(grow) * sizeof(kamd::PathArc)));
    D->path_cap = grow;
  }
  int *d_n = static_cast<int *>(D->d_path); float *d_fc = reinterpret_cast<float *>(d_n + 1);
  kamd::PathArc *d_arcs = reinterpret_cast<kamd::PathArc *>(static_cast<char *>(D->d_path) + 16);
  hipStream_t st = D->last_stream;
  hipLaunchKernelGGL(kamd::TracebackKernel, dim3(1), dim3(NT), 0, st, D->dev, lane, use_final_probs, d_arcs, cap, d_n, d_fc);
  KAMD_HIP(hipGetLastError());
  struct { int n; float fc; } head = {0, 0.f};
  KAMD_HIP(hipMemcpyAsync(&head, D->d_path, sizeof(head), hipMemcpyDeviceToHost, st));
  KAMD_HIP(hipStreamSynchronize(st));
  const int n = head.n; const float fc = head.fc;
  std::vector<kamd::PathArc> arcs(n > 0 ? std::min(n, cap) : 0);
  if (!arcs.empty()) {
    KAMD_HIP(hipMemcpyAsync(arcs.data(), d_arcs, arcs.size() * sizeof(kamd::PathArc), hipMemcpyDeviceToHost, st));
    KAMD_HIP(hipStreamSynchronize(st));
  }
  *ali_len = 0; *words_len = 0; *graph_cost = INFINITY; *acoustic_cost = INFINITY;
  if (n < 0) return kamd::SetError(KAMD_ERR_STATE, "no tokens alive on the newest frame");
  float g = 0.f, a = 0.f;   // Times() along the path, start -> end (fstext/lattice-weight.h)
  for (int i = static_cast<int>(arcs.size()) - 1; i >= 0; i--) {
    if (arcs[i].ilabel != 0) { if (*ali_len < ali_cap) alignment[*ali_len] = arcs[i].ilabel; (*ali_len)++; }
    if (arcs[i].olabel != 0) { if (*words_len < words_cap) words[*words_len] = arcs[i].olabel; (*words_len)++; }
    g += arcs[i].graph; a += arcs[i].ac;
  }
  *graph_cost = g + fc; *acoustic_cost = a;
  return KAMD_OK;
}

// ---- endpointing (online2/online-endpoint.{h,cc})
void kamd_endpoint_config_default(kamd_endpoint_config *c) {
  // OnlineEndpointConfig(), online2/online-endpoint.h:149-154
  const kamd_endpoint_rule r[5] = {{0, 5.0f, INFINITY, 0.0f}, {1, 0.5f, 2.0f, 0.0f}, {1, 1.0f, 8.0f, 0.0f},
                                   {1, 2.0f, INFINITY, 0.0f}, {0, 0.0f, INFINITY, 20.0f}};
  for (int i = 0; i < 5; i++) c->rule[i] = r[i];
}

int kamd_endpoint_detected(const kamd_endpoint_config *c, int num_frames_decoded, int trailing_silence_frames,
                           float frame_shift_in_seconds, float final_relative_cost) {
  if (!c || num_frames_decoded < trailing_silence_frames || trailing_silence_frames < 0)
  {
    kamd::SetError(KAMD_ERR_ARG, "endpointing: %d frames decoded, %d trailing silence frames", num_frames_decoded, trailing_silence_frames);
    return -1;
  }
  const float utterance_length = num_frames_decoded * frame_shift_in_seconds,
              trailing_silence = trailing_silence_frames * frame_shift_in_seconds;
  const bool contains_nonsilence = utterance_length > trailing_silence;
  for (int i = 0; i < 5; i++) {   // RuleActivated, online-endpoint.cc:25-44
    const kamd_endpoint_rule &r = c->rule[i];
    if ((contains_nonsilence || !r.must_contain_nonsilence) && trailing_silence >= r.min_trailing_silence &&
        final_relative_cost <= r.max_relative_cost && utterance_length >= r.min_utterance_length)
      return 1;
  }
  return 0;
}

int kamd_decoder_set_silence_phones(kamd_decoder *h, const int32_t *tid2phone, int32_t num_tids,
                                    const int32_t *silence_phones, int n_sil) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (!D || !tid2phone || num_tids <= 0) return kamd::SetError(KAMD_ERR_ARG, "endpointing needs the transition-id -> phone table");
  // "Endpointing requires nonempty --endpoint.silence-phones option"; duplicates are an error too (:77-82)
  if (!silence_phones || n_sil <= 0) return kamd::SetError(KAMD_ERR_ARG, "Endpointing requires nonempty --endpoint.silence-phones option");
  std::vector<int32_t> sp(silence_phones, silence_phones + n_sil);
  std::sort(sp.begin(), sp.end());
  if (std::adjacent_find(sp.begin(), sp.end()) != sp.end())
    return kamd::SetError(KAMD_ERR_ARG, "Duplicates in --silence-phones option in endpointing config");
  std::vector<unsigned char> tbl(static_cast<size_t>(num_tids) + 1, 0);
  for (int t = 1; t <= num_tids; t++) tbl[t] = std::binary_search(sp.begin(), sp.end(), tid2phone[t]) ? 1 : 0;
  if (D->d_sil_tid) { (void)hipFree(D->d_sil_tid); D->d_sil_tid = NULL; }
  KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_sil_tid), tbl.size()));
  KAMD_HIP(hipMemcpy(D->d_sil_tid, tbl.data(), tbl.size(), hipMemcpyHostToDevice));
  D->n_sil_tids = num_tids + 1;
  return KAMD_OK;
}

int kamd_decoder_trailing_silence_frames(kamd_decoder *h, const int32_t *lanes, int n, int32_t *out) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (!D->d_sil_tid) return kamd::SetError(KAMD_ERR_STATE, "kamd_decoder_set_silence_phones has not been called");
  if (CheckLanes(D, lanes, n) != KAMD_OK || EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_ARG;
  const int rc = kamd_decoder_sync(h);
  if (rc != KAMD_OK) return rc;
  for (int i = 0; i < n; i++)
    if (D->h_st[lanes[i]].finalized)   // BestPathEnd: "decoding_finalized_ && !use_final_probs" is an error (:84-87)
      return kamd::SetError(KAMD_ERR_STATE, "lane %d is finalized: no trailing-silence traceback without final-probs", lanes[i]);
  if (n > D->sil_out_cap) {
    if (D->d_sil_out) (void)hipFree(D->d_sil_out);
    D->d_sil_out = NULL; D->sil_out_cap = 0;
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_sil_out), std::max(n, 64) * sizeof(int)));
    D->sil_out_cap = std::max(n, 64);
  }
  hipStream_t st = D->last_stream;
  KAMD_HIP(hipMemcpyAsync(D->d_lanes, lanes, n * sizeof(int), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(kamd::TrailingSilenceKernel, dim3(n), dim3(NT), 0, st, D->dev, D->d_lanes, D->d_sil_tid, D->n_sil_tids, D->d_sil_out);
  KAMD_HIP(hipGetLastError());
  KAMD_HIP(hipMemcpyAsync(out, D->d_sil_out, n * sizeof(int), hipMemcpyDeviceToHost, st));
  KAMD_HIP(hipStreamSynchronize(st));
  return KAMD_OK;
}

int kamd_decoder_endpoint_detected(kamd_decoder *h, const kamd_endpoint_config *cfg, const int32_t *lanes, int n,
                                   float frame_shift_in_seconds, int32_t *detected, int32_t *trailing_silence_frames) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (!cfg || !detected) return kamd::SetError(KAMD_ERR_ARG, "endpointing: null argument");
  std::vector<int32_t> sil(n, 0);
  // (syncs: FinalRelativeCost / NumFramesDecoded below are those of the newest frame)
  const int rc = kamd_decoder_trailing_silence_frames(h, lanes, n, sil.data());
  if (rc != KAMD_OK) return rc;
  for (int i = 0; i < n; i++) {
    const kamd::LaneState &S = D->h_st[lanes[i]];
    // EndpointDetected(config, tmodel, shift, decoder): false before the first frame (:110)
    detected[i] = S.frame == 0 ? 0 : kamd_endpoint_detected(cfg, S.frame, sil[i], frame_shift_in_seconds, S.final_relative_cost);
    if (detected[i] < 0) return KAMD_ERR_ARG;
    if (trailing_silence_frames) trailing_silence_frames[i] = sil[i];
  }
  return KAMD_OK;
}

int kamd_decoder_partial_best_paths(kamd_decoder *h, const int32_t *lanes, int n, int use_final_probs, int32_t *alignments, int ali_cap,
                                    int32_t *ali_len, int32_t *words, int words_cap, int32_t *words_len, float *graph_cost,
                                    float *acoustic_cost) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (CheckLanes(D, lanes, n) != KAMD_OK || EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_ARG;
  int max_frame = 0;
  for (int i = 0; i < n; i++) {
    if (D->h_st[lanes[i]].finalized) return kamd::SetError(KAMD_ERR_STATE, "lane %d is finalized: use kamd_decoder_best_path", lanes[i]);
    max_frame = std::max(max_frame, D->h_st[lanes[i]].frame);
  }
  const int cap = 4 * (max_frame + 2) + 1024;
  const size_t arcs_bytes = static_cast<size_t>(n) * cap * sizeof(kamd::PathArc), head_bytes = static_cast<size_t>(n) * 8;
  // grow-only buffer owned by the decoder (a server asks for partial results every tick: no hipMalloc / hipFree,
  // both of which synchronise the device, on that path)
  if (arcs_bytes + head_bytes > D->paths_cap) {
    const size_t grow = std::max(arcs_bytes + head_bytes, 2 * D->paths_cap);
    if (D->d_paths) (void)hipFree(D->d_paths);
    D->d_paths = NULL; D->paths_cap = 0;
    KAMD_HIP(hipMalloc(&D->d_paths, grow));
    D->paths_cap = grow;
  }
  void *d_buf = D->d_paths;
  kamd::PathArc *d_arcs = static_cast<kamd::PathArc *>(d_buf);
  int *d_head = reinterpret_cast<int *>(static_cast<char *>(d_buf) + arcs_bytes);
  hipStream_t st = D->last_stream;
  std::vector<int> head(2 * static_cast<size_t>(n));
  std::vector<kamd::PathArc> arcs;
  int rc = KAMD_OK;
  if (hipMemcpyAsync(D->d_lanes, lanes, n * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
    rc = kamd::SetError(KAMD_ERR_HIP, "traceback: upload failed");
  if (rc == KAMD_OK) {
    hipLaunchKernelGGL(kamd::TracebackBatchKernel, dim3(n), dim3(NT), 0, st, D->dev, D->d_lanes, use_final_probs, d_arcs, cap, d_head);
    if (hipMemcpyAsync(head.data(), d_head, head_bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
      rc = kamd::SetError(KAMD_ERR_HIP, "traceback kernel failed: %s", hipGetErrorString(hipGetLastError()));
  }
  if (rc == KAMD_OK) {
    int longest = 0;
    for (int i = 0; i < n; i++) longest = std::max(longest, std::min(head[2 * i], cap));
    // one strided copy of the used prefix of every lane's segment
    arcs.resize(static_cast<size_t>(n) * std::max(longest, 1));
    if (longest > 0 && hipMemcpy2D(arcs.data(), static_cast<size_t>(longest) * sizeof(kamd::PathArc), d_arcs, static_cast<size_t>(cap) * sizeof(kamd::PathArc),
                                   static_cast<size_t>(longest) * sizeof(kamd::PathArc), n, hipMemcpyDeviceToHost) != hipSuccess)
      rc = kamd::SetError(KAMD_ERR_HIP, "traceback copy failed");
    for (int i = 0; rc == KAMD_OK && i < n; i++) {
      int32_t *ali = alignments + static_cast<size_t>(i) * ali_cap, *wrd = words + static_cast<size_t>(i) * words_cap;
      ali_len[i] = 0; words_len[i] = 0; graph_cost[i] = INFINITY; acoustic_cost[i] = INFINITY;
      const int cnt = head[2 * i];
      if (cnt < 0) { ali_len[i] = -1; words_len[i] = -1; continue; }      // no tokens alive on the newest frame
      if (cnt > cap) {
        // the lane advanced since the last sync (the capacity was sized from the host's copy of its frame count): a path
        // cut to its newest `cap` arcs would come back with wrong costs and words
        rc = kamd::SetError(KAMD_ERR_STATE, "lane %d: best path of %d arcs exceeds the buffer sized from the last kamd_decoder_sync (%d): sync first",
                            lanes[i], cnt, cap);
        break;
      }
      float fc; memcpy(&fc, &head[2 * i + 1], 4);
      const kamd::PathArc *A = arcs.data() + static_cast<size_t>(i) * longest;
      float gsum = 0.f, asum = 0.f;   // Times() along the path, start -> end
      for (int k = std::min(cnt, cap) - 1; k >= 0; k--) {
        if (A[k].ilabel != 0) { if (ali_len[i] < ali_cap) ali[ali_len[i]] = A[k].ilabel; ali_len[i]++; }
        if (A[k].olabel != 0) { if (words_len[i] < words_cap) wrd[words_len[i]] = A[k].olabel; words_len[i]++; }
        gsum += A[k].graph; asum += A[k].ac;
      }
      graph_cost[i] = gsum + fc; acoustic_cost[i] = asum;
    }
  }
  return rc;
}

int kamd_decoder_partial_best_paths_incremental(kamd_decoder *h, const int32_t *lanes, int n, int32_t *alignments, int ali_cap, int32_t *ali_len,
                                                int32_t *words, int words_cap, int32_t *words_len, float *graph_cost, float *acoustic_cost) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return KAMD_OK;
  if (CheckLanes(D, lanes, n) != KAMD_OK || EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_ARG;
  int max_frame = 0;
  for (int i = 0; i < n; i++) {
    if (D->h_st[lanes[i]].finalized) return kamd::SetError(KAMD_ERR_STATE, "lane %d is finalized: use kamd_decoder_best_path", lanes[i]);
    max_frame = std::max(max_frame, D->h_st[lanes[i]].frame);
  }
  if (!D->d_pp_arcs) {
    D->pp_stride = D->sizes.max_frames + 2;
    D->pp_cap = 2 * (4 * (D->sizes.max_frames + 2) + 1024);
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_pp_arcs), static_cast<size_t>(D->sizes.max_lanes) * D->pp_cap * sizeof(kamd::PathArc)));
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_pp_rec), static_cast<size_t>(D->sizes.max_lanes) * 4 * D->pp_stride * sizeof(int)));
    D->pp_known.assign(D->sizes.max_lanes, 0);
  }
  const int cap = 4 * (max_frame + 2) + 1024;
  const size_t arcs_bytes = static_cast<size_t>(n) * cap * sizeof(kamd::PathArc), head_bytes = static_cast<size_t>(n) * 12, known_bytes = static_cast<size_t>(n) * 4;
  if (arcs_bytes + head_bytes + known_bytes > D->paths_cap) {
    const size_t grow = std::max(arcs_bytes + head_bytes + known_bytes, 2 * D->paths_cap);
    if (D->d_paths) (void)hipFree(D->d_paths);
    D->d_paths = NULL; D->paths_cap = 0;
    KAMD_HIP(hipMalloc(&D->d_paths, grow));
    D->paths_cap = grow;
  }
  kamd::PathArc *d_arcs = static_cast<kamd::PathArc *>(D->d_paths);
  int *d_head = reinterpret_cast<int *>(static_cast<char *>(D->d_paths) + arcs_bytes);
  int *d_known = d_head + 3 * n;
  hipStream_t st = D->last_stream;
  std::vector<int> known(n), head(3 * static_cast<size_t>(n));
  for (int i = 0; i < n; i++) known[i] = D->pp_known[lanes[i]];
  KAMD_HIP(hipMemcpyAsync(d_known, known.data(), known_bytes, hipMemcpyHostToDevice, st));
  KAMD_HIP(hipMemcpyAsync(D->d_lanes, lanes, n * sizeof(int), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(kamd::TracebackIncKernel, dim3(n), dim3(NT), 0, st, D->dev, D->d_lanes, d_known, D->d_pp_arcs, D->pp_cap, D->d_pp_rec, D->pp_stride,
                     d_arcs, cap, d_head);
  KAMD_HIP(hipMemcpyAsync(head.data(), d_head, head_bytes, hipMemcpyDeviceToHost, st));
  if (hipStreamSynchronize(st) != hipSuccess) return kamd::SetError(KAMD_ERR_HIP, "traceback kernel failed: %s", hipGetErrorString(hipGetLastError()));
  int longest = 0;
  for (int i = 0; i < n; i++) {
    if (head[3 * i] == -2)
      return kamd::SetError(KAMD_ERR_STATE, "lane %d: the best path exceeds the buffer sized from the last kamd_decoder_sync (%d arcs): sync first", lanes[i], cap);
    longest = std::max(longest, head[3 * i]);
  }
  std::vector<kamd::PathArc> arcs(static_cast<size_t>(n) * std::max(longest, 1));
  if (longest > 0)      // one strided copy of the used prefix of every lane's segment
    KAMD_HIP(hipMemcpy2D(arcs.data(), static_cast<size_t>(longest) * sizeof(kamd::PathArc), d_arcs, static_cast<size_t>(cap) * sizeof(kamd::PathArc),
                         static_cast<size_t>(longest) * sizeof(kamd::PathArc), n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    int32_t *ali = alignments + static_cast<size_t>(i) * ali_cap, *wrd = words + static_cast<size_t>(i) * words_cap;
    ali_len[i] = 0; words_len[i] = 0; graph_cost[i] = INFINITY; acoustic_cost[i] = INFINITY;
    const int cnt = head[3 * i];
    if (cnt < 0) { ali_len[i] = -1; words_len[i] = -1; D->pp_known[lanes[i]] = 0; continue; }      // no tokens alive on the newest frame
    float fc; memcpy(&fc, &head[3 * i + 1], 4);
    const kamd::PathArc *A = arcs.data() + static_cast<size_t>(i) * longest;
    float gsum = 0.f, asum = 0.f;   // Times() along the path, start -> end (the arcs come oldest first here)
    for (int k = 0; k < cnt; k++) {
      if (A[k].ilabel != 0) { if (ali_len[i] < ali_cap) ali[ali_len[i]] = A[k].ilabel; ali_len[i]++; }
      if (A[k].olabel != 0) { if (words_len[i] < words_cap) wrd[words_len[i]] = A[k].olabel; words_len[i]++; }
      gsum += A[k].graph; asum += A[k].ac;
    }
    graph_cost[i] = gsum + fc; acoustic_cost[i] = asum;
    D->pp_known[lanes[i]] = head[3 * i + 2];
  }
  return KAMD_OK;
}

static int FrameTracebacks(Decoder *D, const int32_t *lanes, int n, int incremental, int32_t *tids, int32_t *tokens, int cap,
                           int32_t *n_decoded, int32_t *n_entries) {
  if (n <= 0) return KAMD_OK;
  if (CheckLanes(D, lanes, n) != KAMD_OK || EnsureTaskBuf(D, n) != KAMD_OK) return KAMD_ERR_ARG;
  int max_frame = 0;
  for (int i = 0; i < n; i++) {
    if (D->h_st[lanes[i]].finalized) return kamd::SetError(KAMD_ERR_STATE, "lane %d is finalized", lanes[i]);
    max_frame = std::max(max_frame, D->h_st[lanes[i]].frame);
  }
  const int dcap = max_frame + 1;
  const size_t pair_bytes = static_cast<size_t>(n) * dcap * sizeof(int2), head_bytes = static_cast<size_t>(n) * 8, known_bytes = static_cast<size_t>(n) * 4;
  if (pair_bytes + head_bytes + known_bytes > D->paths_cap) {
    const size_t grow = std::max(pair_bytes + head_bytes + known_bytes, 2 * D->paths_cap);
    if (D->d_paths) (void)hipFree(D->d_paths);
    D->d_paths = NULL; D->paths_cap = 0;
    KAMD_HIP(hipMalloc(&D->d_paths, grow));
    D->paths_cap = grow;
  }
  int2 *d_pairs = static_cast<int2 *>(D->d_paths);
  int *d_head = reinterpret_cast<int *>(static_cast<char *>(D->d_paths) + pair_bytes);
  int *d_known = d_head + 2 * n;
  hipStream_t st = D->last_stream;
  std::vector<int> known(n, 0);
  if (incremental) {
    if (!D->d_trace_tok) {
      D->trace_stride = D->sizes.max_frames + 2;
      KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_trace_tok), static_cast<size_t>(D->sizes.max_lanes) * D->trace_stride * sizeof(int)));
      D->trace_known.assign(D->sizes.max_lanes, 0);
    }
    for (int i = 0; i < n; i++) known[i] = D->trace_known[lanes[i]];
    KAMD_HIP(hipMemcpyAsync(d_known, known.data(), known_bytes, hipMemcpyHostToDevice, st));
  }
  KAMD_HIP(hipMemcpyAsync(D->d_lanes, lanes, n * sizeof(int), hipMemcpyHostToDevice, st));
  KAMD_HIP(hipStreamSynchronize(st));
  hipLaunchKernelGGL(kamd::FrameTraceKernel, dim3(n), dim3(NT), 0, st, D->dev, D->d_lanes, d_known, incremental ? D->d_trace_tok : NULL,
                     D->trace_stride, d_pairs, dcap, d_head);
  std::vector<int> head(2 * static_cast<size_t>(n));
  KAMD_HIP(hipMemcpyAsync(head.data(), d_head, head_bytes, hipMemcpyDeviceToHost, st));
  if (hipStreamSynchronize(st) != hipSuccess) return kamd::SetError(KAMD_ERR_HIP, "traceback kernel failed: %s", hipGetErrorString(hipGetLastError()));
  int longest = 0;
  for (int i = 0; i < n; i++) {
    if (head[2 * i + 1] > dcap)
      return kamd::SetError(KAMD_ERR_STATE, "lane %d advanced since the last kamd_decoder_sync (%d frames on the path, %d known): sync first", lanes[i],
                            head[2 * i + 1], dcap);
    if (head[2 * i + 1] > cap) return kamd::SetError(KAMD_ERR_ARG, "lane %d: %d frames to report, room for %d", lanes[i], head[2 * i + 1], cap);
    longest = std::max(longest, head[2 * i + 1]);
  }
  std::vector<int2> pairs(static_cast<size_t>(n) * std::max(longest, 1));
  if (longest > 0)      // one strided copy of the used prefix of every lane's segment
    KAMD_HIP(hipMemcpy2D(pairs.data(), static_cast<size_t>(longest) * sizeof(int2), d_pairs, static_cast<size_t>(dcap) * sizeof(int2),
                         static_cast<size_t>(longest) * sizeof(int2), n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; i++) {
    n_decoded[i] = head[2 * i];
    const int m = head[2 * i] < 0 ? 0 : head[2 * i + 1];
    if (n_entries) n_entries[i] = m;
    for (int k = 0; k < m; k++) {
      tids[static_cast<size_t>(i) * cap + k] = pairs[static_cast<size_t>(i) * longest + k].x;
      tokens[static_cast<size_t>(i) * cap + k] = pairs[static_cast<size_t>(i) * longest + k].y;
    }
    if (incremental && head[2 * i] >= 0) D->trace_known[lanes[i]] = std::min(head[2 * i], D->trace_stride);
  }
  return KAMD_OK;
}

int kamd_decoder_frame_tracebacks(kamd_decoder *h, const int32_t *lanes, int n, int32_t *tids, int32_t *tokens, int cap, int32_t *counts) {
  return FrameTracebacks(reinterpret_cast<Decoder *>(h), lanes, n, 0, tids, tokens, cap, counts, NULL);
}

int kamd_decoder_frame_tracebacks_incremental(kamd_decoder *h, const int32_t *lanes, int n, int32_t *tids, int32_t *tokens, int cap,
                                              int32_t *n_decoded, int32_t *n_entries) {
  return FrameTracebacks(reinterpret_cast<Decoder *>(h), lanes, n, 1, tids, tokens, cap, n_decoded, n_entries);
}

int kamd_decoder_get_phase_cycles(kamd_decoder *h, int lane, uint64_t cycles[16]) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (LaneOk(D, lane) != KAMD_OK) return KAMD_ERR_ARG;
  for (int i = 0; i < 16; i++) cycles[i] = D->h_st[lane].phase_cycles[i];
  return KAMD_OK;
}
int kamd_decoder_get_trace(kamd_decoder *h, int lane, int32_t *ntok, float *cutoff, float *cost_offset, int cap) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (LaneOk(D, lane) != KAMD_OK) return KAMD_ERR_ARG;
  int n = std::min(cap, D->h_st[lane].frame);
  size_t mf = D->sizes.max_frames;
  if (n > 0) {
    KAMD_HIP(hipMemcpy(ntok, D->dev.trace_ntok + lane * (mf + 1), n * 4, hipMemcpyDeviceToHost));
    KAMD_HIP(hipMemcpy(cutoff, D->dev.trace_cutoff + lane * (mf + 1), n * 4, hipMemcpyDeviceToHost));
    KAMD_HIP(hipMemcpy(cost_offset, D->dev.cost_offsets + lane * (mf + 1), n * 4, hipMemcpyDeviceToHost));
  }
  return n;
}

// GetRawLattice (lattice-faster-decoder.cc:113-196): canonical numbering of the surviving tokens / links.
// st / co: [nt] HCLG state and forward cost in arena order (frame by frame, toff[f] = first token of frame f);
// last_final: final cost of every token of frame F in arena order; lk: links whose endpoints are indices
// into st (after subtracting link_index_base).
using kamd::RawLat;
static int Canonicalize(int nt, int nl, int F, const int *st, const float *co, const int *toff, const float *last_final,
                        const kamd::Link *lk, int link_index_base, int graph_start, RawLat *out) {
  std::vector<float> fin(nt, INFINITY);
  std::vector<int32_t> lat_frame(nt, 0);
  for (int f = 0; f <= F; f++)
    for (int i = toff[f]; i < toff[f + 1] && i < nt; i++) lat_frame[i] = f;
  // every frame must keep a token or the reference produces no lattice (:145-149)
  bool empty_frame = false;
  for (int f = 0; f <= F; f++) if (toff[f + 1] <= toff[f]) empty_frame = true;
  {
    const int lb = toff[F], le = std::min(toff[F + 1], nt);
    bool any = false;
    for (int i = lb; i < le; i++) if (last_final[i - lb] != INFINITY) any = true;
    // :183-192: final weight = final_cost if any final token exists, else One()
    // (any is evaluated over ALL last-frame tokens in the reference, i.e. before pruning;
    //  a surviving non-final set with finals_empty false cannot occur: non-final tokens
    //  get extra_cost = +inf on the last frame when finals exist.)
    for (int i = lb; i < le; i++) fin[i] = any ? last_final[i - lb] : 0.0f;
  }
  // canonical numbering: by (frame, HCLG state)
  std::vector<int> order(nt), inv(nt);
  for (int i = 0; i < nt; i++) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) {
    if (lat_frame[a] != lat_frame[b]) return lat_frame[a] < lat_frame[b];
    return st[a] < st[b];
  });
  for (int i = 0; i < nt; i++) inv[order[i]] = i;
  out->frame.resize(nt); out->hclg.resize(nt); out->cost.resize(nt); out->fin.resize(nt);
  for (int i = 0; i < nt; i++) { out->frame[i] = lat_frame[order[i]]; out->hclg[i] = st[order[i]]; out->cost[i] = co[order[i]]; out->fin[i] = fin[order[i]]; }
  out->arcs.resize(nl);
  for (int i = 0; i < nl; i++) {
    kamd_lat_arc a;
    const int ls = lk[i].src - link_index_base, ld = lk[i].dst - link_index_base;
    if (ls < 0 || ls >= nt || ld < 0 || ld >= nt) return kamd::SetError(KAMD_ERR_STATE, "lattice link %d out of range", i);
    a.src = inv[ls]; a.dst = inv[ld]; a.ilabel = lk[i].ilabel; a.olabel = lk[i].olabel;
    a.graph_cost = lk[i].graph; a.acoustic_cost = lk[i].ac;
    out->arcs[i] = a;
  }
  std::sort(out->arcs.begin(), out->arcs.end(), [](const kamd_lat_arc &a, const kamd_lat_arc &b) {
    if (a.src != b.src) return a.src < b.src;
    if (a.dst != b.dst) return a.dst < b.dst;
    if (a.ilabel != b.ilabel) return a.ilabel < b.ilabel;
    if (a.olabel != b.olabel) return a.olabel < b.olabel;
    if (a.graph_cost != b.graph_cost) return a.graph_cost < b.graph_cost;
    return a.acoustic_cost < b.acoustic_cost;
  });
  out->start = -1;
  for (int i = 0; i < nt && out->frame[i] == 0; i++)
    if (out->hclg[i] == graph_start) out->start = i;
  out->frames = F;
  if (empty_frame) { out->frame.clear(); out->hclg.clear(); out->cost.clear(); out->fin.clear(); out->arcs.clear(); out->start = -1; }
  return KAMD_OK;
}

static int FetchLattice(Decoder *D, int lane) {
  if (D->cached_lane == lane) return KAMD_OK;
  if (LaneOk(D, lane) != KAMD_OK) return KAMD_ERR_ARG;
  const kamd::LaneState &S = D->h_st[lane];
  if (!S.finalized) return kamd::SetError(KAMD_ERR_STATE, "lane %d: call kamd_decoder_finalize + kamd_decoder_sync first", lane);
  const int nt = S.out_ntok, nl = S.out_nlink, F = S.frame;
  const size_t mf = D->sizes.max_frames;
  const long long tbase = D->h_tok_base[lane] + S.out_tok_base, lbase = D->h_lnk_base[lane] + S.out_lnk_base;
  std::vector<int> st(nt), toff(F + 2);
  std::vector<float> co(nt);
  std::vector<kamd::Link> lk(nl);
  if (nt) {
    KAMD_HIP(hipMemcpy(st.data(), D->dev.tok_state + tbase, nt * 4, hipMemcpyDeviceToHost));
    const void *cost_src = S.out_cost_in_map ? static_cast<const void *>(D->dev.tok_map + tbase)
                                             : static_cast<const void *>(D->dev.tok_cost + tbase);
    KAMD_HIP(hipMemcpy(co.data(), cost_src, nt * 4, hipMemcpyDeviceToHost));
  }
  KAMD_HIP(hipMemcpy(toff.data(), D->dev.tok_off + lane * (mf + 2), (F + 2) * 4, hipMemcpyDeviceToHost));
  if (nl) KAMD_HIP(hipMemcpy(lk.data(), D->dev.links + lbase, nl * sizeof(kamd::Link), hipMemcpyDeviceToHost));
  // final costs of the last frame's states
  const int lb = toff[F], le = std::min(toff[F + 1], nt);
  std::vector<float> fc(std::max(0, le - lb));
  // (from the host copy of the graph's final costs made by kamd_graph_create: until round 6 this was one synchronous 4-byte
  // hipMemcpy per token of the last frame -- the end-of-utterance latency of a streaming host)
  const std::vector<float> &gfin = D->g->h_final;
  for (int i = lb; i < le; i++) {
    if (st[i] < 0 || static_cast<size_t>(st[i]) >= gfin.size()) return kamd::SetError(KAMD_ERR_STATE, "lattice token %d: state %d out of range", i, st[i]);
    fc[i - lb] = gfin[st[i]];
  }
  RawLat R;
  const int rc = Canonicalize(nt, nl, F, st.data(), co.data(), toff.data(), fc.data(), lk.data(), S.out_tok_base, D->g->dev.start, &R);
  if (rc != KAMD_OK) return rc;
  D->lat_frame.swap(R.frame); D->lat_hclg.swap(R.hclg); D->lat_cost.swap(R.cost); D->lat_final.swap(R.fin);
  D->lat_arcs.swap(R.arcs); D->lat_start = R.start; D->lat_frames = R.frames;
  D->cached_lane = lane;
  return KAMD_OK;
}

// GetRawLattice on a LIVE decoder (lattice-faster-decoder.cc:113-196 with !decoding_finalized_: every token and forward
// link the decoder holds, final costs computed on the spot): what SingleUtteranceNnet3Decoder::GetLattice(end_of_utterance
// = false) reads between two chunks (online2/online-nnet3-decoding.cc:66-79).  The lane's arenas ARE that lattice --
// frame f's tokens [tok_off[f], tok_off[f + 1]), the emitting links into frame f [lnk_off[2 f], lnk_off[2 f + 1]) with the
// frame's cost offset still inside their acoustic cost (:173-177), its epsilon links behind them; recorded arcs that did
// not pass the frame's final cutoff carry a negative endpoint -- so this is a host-side read of the arenas, nothing
// is launched and the decoder goes on afterwards.  use_final_probs = 0: every token of the last frame is final with
// weight One (:183-192 with use_final_probs false).
static int FetchLiveLattice(Decoder *D, int lane, int use_final_probs, RawLat *R) {
  if (LaneOk(D, lane) != KAMD_OK) return KAMD_ERR_ARG;
  const kamd::LaneState &S = D->h_st[lane];
  if (S.finalized) return kamd::SetError(KAMD_ERR_STATE, "lane %d is finalized: use kamd_decoder_get_raw_lattice", lane);
  const int F = S.frame;
  const size_t mf = D->sizes.max_frames;
  std::vector<int> toff(F + 2), loff(2 * F + 3);
  KAMD_HIP(hipMemcpy(toff.data(), D->dev.tok_off + lane * (mf + 2), (F + 2) * 4, hipMemcpyDeviceToHost));
  KAMD_HIP(hipMemcpy(loff.data(), D->dev.lnk_off + lane * (2 * (mf + 2) + 1), (2 * F + 3) * 4, hipMemcpyDeviceToHost));
  const int nt = toff[F + 1], nl_all = loff[2 * F + 2];
  if (nt <= 0) { *R = RawLat(); R->frames = F; return KAMD_OK; }
  const long long tbase = D->h_tok_base[lane], lbase = D->h_lnk_base[lane];
  std::vector<int> st(nt);
  std::vector<float> co(nt), cof(std::max(F, 1));
  std::vector<kamd::Link> all(std::max(nl_all, 1)), lk;
  KAMD_HIP(hipMemcpy(st.data(), D->dev.tok_state + tbase, nt * 4, hipMemcpyDeviceToHost));
  KAMD_HIP(hipMemcpy(co.data(), D->dev.tok_cost + tbase, nt * 4, hipMemcpyDeviceToHost));
  if (F > 0) KAMD_HIP(hipMemcpy(cof.data(), D->dev.cost_offsets + lane * (mf + 1), F * 4, hipMemcpyDeviceToHost));
  if (nl_all > 0) KAMD_HIP(hipMemcpy(all.data(), D->dev.links + lbase, static_cast<size_t>(nl_all) * sizeof(kamd::Link), hipMemcpyDeviceToHost));
  lk.reserve(nl_all);
  for (int f = 0; f <= F; f++)
    for (int part = 0; part < 2; part++) {
      const float off = (part == 0 && f > 0) ? cof[f - 1] : 0.0f;
      for (int i = loff[2 * f + part]; i < loff[2 * f + part + 1] && i < nl_all; i++) {
        kamd::Link L = all[i];
        if (L.src < 0 || L.dst < 0) continue;
        if (part == 0) L.ac = L.ac - off;
        lk.push_back(L);
      }
    }
  const int lb = toff[F], le = nt;
  std::vector<float> fc(std::max(0, le - lb), 0.0f);
  if (use_final_probs && le > lb) {
    const std::vector<float> &gfin = D->g->h_final;   // host copy made by kamd_graph_create: the graph is shared, read-only from here on
    for (int i = lb; i < le; i++) fc[i - lb] = gfin[st[i]];
  }
  return Canonicalize(nt, static_cast<int>(lk.size()), F, st.data(), co.data(), toff.data(), fc.data(), lk.data(), 0, D->g->dev.start, R);
}

int kamd_decoder_live_lattice_size(kamd_decoder *h, int lane, int use_final_probs, kamd_lattice_size *sz) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  RawLat R;
  const int rc = FetchLiveLattice(D, lane, use_final_probs, &R);
  if (rc != KAMD_OK) return rc;
  sz->num_states = static_cast<int32_t>(R.frame.size()); sz->num_arcs = static_cast<int32_t>(R.arcs.size());
  sz->num_frames = R.frames; sz->start = R.start;
  D->live.frame.swap(R.frame); D->live.hclg.swap(R.hclg); D->live.cost.swap(R.cost); D->live.fin.swap(R.fin); D->live.arcs.swap(R.arcs);
  D->live.start = R.start; D->live.frames = R.frames; D->live_lane = lane; D->live_ufp = use_final_probs;
  return KAMD_OK;
}

int kamd_decoder_get_live_raw_lattice(kamd_decoder *h, int lane, int use_final_probs, int32_t *state_frame, int32_t *state_hclg,
                                      float *state_cost, float *state_final, kamd_lat_arc *arcs) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (D->live_lane != lane || D->live_ufp != use_final_probs || D->live.frames != D->h_st[lane].frame) {
    kamd_lattice_size sz;
    const int rc = kamd_decoder_live_lattice_size(h, lane, use_final_probs, &sz);
    if (rc != KAMD_OK) return rc;
  }
  const RawLat &R = D->live;
  const size_t n = R.frame.size();
  if (n) {
    memcpy(state_frame, R.frame.data(), n * 4); memcpy(state_hclg, R.hclg.data(), n * 4);
    memcpy(state_cost, R.cost.data(), n * 4); memcpy(state_final, R.fin.data(), n * 4);
  }
  if (!R.arcs.empty()) memcpy(arcs, R.arcs.data(), R.arcs.size() * sizeof(kamd_lat_arc));
  D->live_lane = -1;
  return KAMD_OK;
}

int kamd_decoder_lattice_size(kamd_decoder *h, int lane, kamd_lattice_size *sz) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  int rc = FetchLattice(D, lane);
  if (rc != KAMD_OK) return rc;
  sz->num_states = static_cast<int32_t>(D->lat_frame.size());
  sz->num_arcs = static_cast<int32_t>(D->lat_arcs.size());
  sz->num_frames = D->lat_frames; sz->start = D->lat_start;
  return KAMD_OK;
}

int kamd_decoder_get_raw_lattice(kamd_decoder *h, int lane, int32_t *state_frame, int32_t *state_hclg,
                                 float *state_cost, float *state_final, kamd_lat_arc *arcs) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  int rc = FetchLattice(D, lane);
  if (rc != KAMD_OK) return rc;
  size_t n = D->lat_frame.size();
  if (n) {
    memcpy(state_frame, D->lat_frame.data(), n * 4); memcpy(state_hclg, D->lat_hclg.data(), n * 4);
    memcpy(state_cost, D->lat_cost.data(), n * 4); memcpy(state_final, D->lat_final.data(), n * 4);
  }
  if (!D->lat_arcs.empty()) memcpy(arcs, D->lat_arcs.data(), D->lat_arcs.size() * sizeof(kamd_lat_arc));
  return KAMD_OK;
}

// fstext/lattice-weight.h Compare: by value1+value2, then value1
static inline bool LatBetter(float a1, float a2, float b1, float b2) {
  float fa = a1 + a2, fb = b1 + b2;
  if (fa < fb) return true;
  if (fa > fb) return false;
  return a1 < b1;
}

int kamd_lattice_best_path(int32_t n, int32_t start, const float *state_final, const kamd_lat_arc *A, int32_t m,
                           int32_t *alignment, int ali_cap, int *ali_len, int32_t *words, int words_cap, int *words_len,
                           float *graph_cost, float *acoustic_cost) {
  *ali_len = 0; *words_len = 0; *graph_cost = INFINITY; *acoustic_cost = INFINITY;
  if (n <= 0 || start < 0 || start >= n) return kamd::SetError(KAMD_ERR_STATE, "empty lattice");
  // arcs need not be sorted: CSR by source
  std::vector<int> first(n + 1, 0), indeg(n, 0), by_src(m);
  for (int i = 0; i < m; i++) {
    if (A[i].src < 0 || A[i].src >= n || A[i].dst < 0 || A[i].dst >= n) return kamd::SetError(KAMD_ERR_ARG, "lattice arc %d out of range", i);
    first[A[i].src + 1]++; indeg[A[i].dst]++;
  }
  for (int s = 0; s < n; s++) first[s + 1] += first[s];
  {
    std::vector<int> fill(first.begin(), first.end() - 1);
    for (int i = 0; i < m; i++) by_src[fill[A[i].src]++] = i;
  }
  std::vector<float> d1(n, INFINITY), d2(n, INFINITY);
  std::vector<int> back(n, -1), stack;
  for (int s = n - 1; s >= 0; s--) if (indeg[s] == 0) stack.push_back(s);
  d1[start] = 0.0f; d2[start] = 0.0f;
  size_t visited = 0;
  while (!stack.empty()) {
    int s = stack.back(); stack.pop_back(); visited++;
    for (int k = first[s]; k < first[s + 1]; k++) {
      const kamd_lat_arc &a = A[by_src[k]];
      if (d1[s] != INFINITY) {
        float n1 = d1[s] + a.graph_cost, n2 = d2[s] + a.acoustic_cost;
        if (d1[a.dst] == INFINITY || LatBetter(n1, n2, d1[a.dst], d2[a.dst])) { d1[a.dst] = n1; d2[a.dst] = n2; back[a.dst] = by_src[k]; }
      }
      if (--indeg[a.dst] == 0) stack.push_back(a.dst);
    }
  }
  if (visited != static_cast<size_t>(n)) return kamd::SetError(KAMD_ERR_STATE, "lattice has a cycle");
  int best = -1; float b1 = INFINITY, b2 = INFINITY;
  for (int s = 0; s < n; s++) {
    if (state_final[s] == INFINITY || d1[s] == INFINITY) continue;
    float t1 = d1[s] + state_final[s], t2 = d2[s];
    if (best == -1 || LatBetter(t1, t2, b1, b2)) { best = s; b1 = t1; b2 = t2; }
  }
  if (best == -1) return kamd::SetError(KAMD_ERR_STATE, "no path to a final lattice state");
  std::vector<int> path;
  for (int s = best; back[s] != -1; s = A[back[s]].src) path.push_back(back[s]);
  std::reverse(path.begin(), path.end());
  for (size_t i = 0; i < path.size(); i++) {
    const kamd_lat_arc &a = A[path[i]];
    if (a.ilabel != 0) { if (*ali_len < ali_cap) alignment[*ali_len] = a.ilabel; (*ali_len)++; }
    if (a.olabel != 0) { if (*words_len < words_cap) words[*words_len] = a.olabel; (*words_len)++; }
  }
  *graph_cost = b1; *acoustic_cost = b2;
  return KAMD_OK;
}

// fst::Prune on a raw lattice (lat/lattice-functions.cc PruneLattice: total weight = graph + acoustic cost): states and
// arcs on no path within `beam` of the best path go.  The exact form of what LatticeFasterOnlineDecoderTpl::
// GetRawLatticePruned (decoder/lattice-faster-online-decoder.cc:168-265) approximates with the extra costs left by the
// last periodic PruneActiveTokens.  state_map[s] = new number or -1, arc_keep[i] = 0 / 1.
int kamd_lattice_prune(int32_t n, int32_t start, const float *state_final, const kamd_lat_arc *A, int32_t m, float beam,
                       int32_t *state_map, uint8_t *arc_keep, int32_t *n_out, int32_t *m_out) {
  *n_out = 0; *m_out = 0;
  for (int s = 0; s < n; s++) state_map[s] = -1;
  for (int i = 0; i < m; i++) arc_keep[i] = 0;
  if (n <= 0 || start < 0 || start >= n) return KAMD_OK;
  std::vector<int> first(n + 1, 0), indeg(n, 0), by_src(m), order;
  for (int i = 0; i < m; i++) {
    if (A[i].src < 0 || A[i].src >= n || A[i].dst < 0 || A[i].dst >= n) return kamd::SetError(KAMD_ERR_ARG, "lattice arc %d out of range", i);
    first[A[i].src + 1]++; indeg[A[i].dst]++;
  }
  for (int s = 0; s < n; s++) first[s + 1] += first[s];
  {
    std::vector<int> fill(first.begin(), first.end() - 1);
    for (int i = 0; i < m; i++) by_src[fill[A[i].src]++] = i;
  }
  order.reserve(n);
  for (int s = 0; s < n; s++) if (indeg[s] == 0) order.push_back(s);
  for (size_t k = 0; k < order.size(); k++)
    for (int j = first[order[k]]; j < first[order[k] + 1]; j++)
      if (--indeg[A[by_src[j]].dst] == 0) order.push_back(A[by_src[j]].dst);
  if (order.size() != static_cast<size_t>(n)) return kamd::SetError(KAMD_ERR_STATE, "lattice has a cycle");
  std::vector<double> fwd(n, INFINITY), bwd(n, INFINITY);
  fwd[start] = 0.0;
  for (int s : order)
    if (fwd[s] != INFINITY)
      for (int j = first[s]; j < first[s + 1]; j++) {
        const kamd_lat_arc &a = A[by_src[j]];
        fwd[a.dst] = std::min(fwd[a.dst], fwd[s] + static_cast<double>(a.graph_cost) + static_cast<double>(a.acoustic_cost));
      }
  for (size_t k = order.size(); k-- > 0;) {
    const int s = order[k];
    double b = state_final[s];
    for (int j = first[s]; j < first[s + 1]; j++) {
      const kamd_lat_arc &a = A[by_src[j]];
      b = std::min(b, static_cast<double>(a.graph_cost) + static_cast<double>(a.acoustic_cost) + bwd[a.dst]);
    }
    bwd[s] = b;
  }
  const double best = bwd[start];
  if (best == INFINITY) return KAMD_OK;
  const double limit = best + static_cast<double>(beam);
  for (int s = 0; s < n; s++) if (fwd[s] + bwd[s] <= limit) state_map[s] = (*n_out)++;
  for (int i = 0; i < m; i++)
    if (fwd[A[i].src] + static_cast<double>(A[i].graph_cost) + static_cast<double>(A[i].acoustic_cost) + bwd[A[i].dst] <= limit) { arc_keep[i] = 1; (*m_out)++; }
  return KAMD_OK;
}

int kamd_decoder_best_path(kamd_decoder *h, int lane, int32_t *alignment, int ali_cap, int *ali_len,
                           int32_t *words, int words_cap, int *words_len, float *graph_cost,
                           float *acoustic_cost) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  int rc = FetchLattice(D, lane);
  if (rc != KAMD_OK) return rc;
  return kamd_lattice_best_path(static_cast<int32_t>(D->lat_frame.size()), D->lat_start, D->lat_final.data(), D->lat_arcs.data(),
                                static_cast<int32_t>(D->lat_arcs.size()), alignment, ali_cap, ali_len, words, words_cap, words_len,
                                graph_cost, acoustic_cost);
}

// ---------------------------------------------------------------- work queue (host)
int kamd_decoder_queue_configure(kamd_decoder *h, int64_t pool_bytes) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (pool_bytes < 4096) return kamd::SetError(KAMD_ERR_ARG, "lattice pool too small");
  if (static_cast<unsigned long long>(pool_bytes) <= D->pool_cap) return KAMD_OK;
  // The pool lives in page-locked host memory and the kernel writes an utterance's lattice straight into it (~50 KB per
  // utterance over the host link, posted stores): a host thread that sees the utterance's status reads the blob where it
  // is.  (A pool in HBM cost one hipMemcpyAsync + hipStreamSynchronize per utterance -- 14 per millisecond at the
  // headline rate, all through the runtime's locks: the host tail took twice as long per utterance on 64 threads as on 32.)
  if (D->h_pool) (void)hipHostFree(D->h_pool);
  D->h_pool = NULL; D->d_pool = NULL; D->pool_cap = 0;
  KAMD_HIP(hipHostMalloc(reinterpret_cast<void **>(&D->h_pool), static_cast<size_t>(pool_bytes), hipHostMallocDefault));
  void *dp = NULL;
  KAMD_HIP(hipHostGetDevicePointer(&dp, D->h_pool, 0));
  D->d_pool = static_cast<unsigned char *>(dp);
  D->pool_cap = static_cast<unsigned long long>(pool_bytes);
  return KAMD_OK;
}

static int QueueLaunch(kamd_decoder *h, const kamd_queue_task *tasks, int n, int resident_lanes, void *stream, bool wide);

int kamd_decoder_queue_launch(kamd_decoder *h, const kamd_queue_task *tasks, int n, int resident_lanes, void *stream) {
  return QueueLaunch(h, tasks, n, resident_lanes, stream, false);
}

// A launch for few utterances that need much room (the second chance of utterances whose lane ran out of token / link
// arena, NnetBatchDecoder): n lanes, one per task, the whole token and link pools split between just these n -- each
// lane's arenas are max_lanes / n times the usual ones.  Nothing of this decoder may be in flight (the split is
// uploaded synchronously); the next ordinary launch restores the uniform split.
int kamd_decoder_queue_launch_wide(kamd_decoder *h, const kamd_queue_task *tasks, int n, void *stream) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0 || n > D->sizes.max_lanes) return kamd::SetError(KAMD_ERR_ARG, "wide queue launch: %d tasks, the decoder has %d lanes", n, D->sizes.max_lanes);
  std::vector<int32_t> share(n, 1);
  if (kamd_decoder_reserve(h, share.data(), n) != KAMD_OK) return KAMD_ERR_HIP;
  return QueueLaunch(h, tasks, n, n, stream, true);
}

int kamd_decoder_max_lanes(const kamd_decoder *h) { return reinterpret_cast<const Decoder *>(h)->sizes.max_lanes; }
int kamd_decoder_max_frames(const kamd_decoder *h) { return reinterpret_cast<const Decoder *>(h)->sizes.max_frames; }

static int QueueLaunch(kamd_decoder *h, const kamd_queue_task *tasks, int n, int resident_lanes, void *stream, bool wide) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (n <= 0) return kamd::SetError(KAMD_ERR_ARG, "empty queue");
  for (int i = 0; i < n; i++) {
    if (tasks[i].n_frames < 0 || tasks[i].n_frames > D->sizes.max_frames)
      return kamd::SetError(KAMD_ERR_ARG, "task %d: %d frames (max_frames %d)", i, tasks[i].n_frames, D->sizes.max_frames);
    if (tasks[i].utt < 0 || tasks[i].utt >= n) return kamd::SetError(KAMD_ERR_ARG, "task %d: utterance index %d outside [0, %d)", i, tasks[i].utt, n);
    if (tasks[i].n_frames > 0 && tasks[i].ld < D->num_pdfs)
      return kamd::SetError(KAMD_ERR_ARG, "task %d: log-likelihood rows of %d columns, the graph's arcs map to pdfs up to %d", i, tasks[i].ld, D->num_pdfs - 1);
  }
  int cus = 0;
  {
    int dev = 0; hipDeviceProp_t prop;
    KAMD_HIP(hipGetDevice(&dev));
    KAMD_HIP(hipGetDeviceProperties(&prop, dev));
    cus = prop.multiProcessorCount;
  }
  int R = resident_lanes > 0 ? resident_lanes : cus * LANES_PER_CU;
  R = std::min(std::min(R, n), D->sizes.max_lanes);
  if (R < 1) return kamd::SetError(KAMD_ERR_ARG, "no resident lanes");
  if (!wide && ReserveUniform(D) != KAMD_OK) return KAMD_ERR_HIP;
  if (!D->d_pool && kamd_decoder_queue_configure(h, 1ll << 30) != KAMD_OK) return KAMD_ERR_HIP;
  if (!D->d_pool_used) {
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_pool_used), 8));
    KAMD_HIP(hipMalloc(reinterpret_cast<void **>(&D->d_qctl), 2 * sizeof(int)));
  }
  if (n > D->ring_cap) {
    if (D->h_results) (void)hipHostFree(D->h_results);
    if (D->h_ring) (void)hipHostFree(D->h_ring);
    D->h_results = NULL; D->h_ring = NULL; D->ring_cap = 0;
    KAMD_HIP(hipHostMalloc(reinterpret_cast<void **>(&D->h_results), static_cast<size_t>(n) * sizeof(kamd_queue_result), hipHostMallocCoherent | hipHostMallocMapped));
    KAMD_HIP(hipHostMalloc(reinterpret_cast<void **>(&D->h_ring), static_cast<size_t>(n) * sizeof(int), hipHostMallocCoherent | hipHostMallocMapped));
    D->ring_cap = n;
  }
  memset(D->h_results, 0, static_cast<size_t>(n) * sizeof(kamd_queue_result));
  memset(D->h_ring, 0, static_cast<size_t>(n) * sizeof(int));
  hipStream_t st = static_cast<hipStream_t>(stream);
  KAMD_HIP(hipMemsetAsync(D->d_pool_used, 0, 8, st));
  KAMD_HIP(hipMemsetAsync(D->d_qctl, 0, 2 * sizeof(int), st));
  // the task list is pulled in by a kernel on `st`: the launch is issued while the acoustic model still runs (no host wait)
  void *d_tasks = NULL;
  if (D->qtasks.Acquire(tasks, static_cast<size_t>(n) * sizeof(kamd_queue_task), &d_tasks, st) != KAMD_OK) return KAMD_ERR_HIP;
  struct Releaser { kamd::MetaRing &m; hipStream_t s; ~Releaser() { (void)m.Release(s); } } releaser{D->qtasks, st};
  kamd::QueueDev q;
  q.tasks = static_cast<const kamd_queue_task *>(d_tasks); q.n_tasks = n; q.head = D->d_qctl; q.done_count = D->d_qctl + 1;
  q.pool = D->d_pool; q.pool_cap = D->pool_cap; q.pool_used = D->d_pool_used;
  void *dp = NULL;
  KAMD_HIP(hipHostGetDevicePointer(&dp, D->h_results, 0));
  q.results = static_cast<kamd_queue_result *>(dp);
  KAMD_HIP(hipHostGetDevicePointer(&dp, D->h_ring, 0));
  q.done_ring = static_cast<int *>(dp);
  const size_t lds = std::max<size_t>(kamd::AdvanceLdsBytes(D->dev.num_pdfs_lds, LDS_TABLE_CAP), FIN_LDS_BYTES);
  KAMD_HIP(hipEventRecord(D->qev[0], st));
  kamd::DecDev dev = D->dev;
  // (a frame's level-2 share is sized from its candidates; an epsilon closure far larger than them can fill it -- flag 1 --
  // and the second chance must not meet the same wall)
  dev.full_level2 = wide ? 1 : 0;
  hipLaunchKernelGGL(kamd::DecodeQueueKernel, dim3(R), dim3(NT), lds, st, dev, q);
  KAMD_HIP(hipGetLastError());
  KAMD_HIP(hipEventRecord(D->qev[1], st));
  D->q_n = n; D->q_next = 0; D->q_lanes = R; D->q_stream = st; D->last_stream = st; D->cached_lane = -1;
  return KAMD_OK;
}

int kamd_decoder_queue_poll(kamd_decoder *h, int32_t *utts, int cap) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  int got = 0;
  while (got < cap && D->q_next < D->q_n) {
    const int v = __atomic_load_n(&D->h_ring[D->q_next], __ATOMIC_ACQUIRE);
    if (v == 0) break;
    utts[got++] = v - 1;
    D->q_next++;
  }
  return got;
}

int kamd_decoder_queue_result(kamd_decoder *h, int32_t utt, kamd_queue_result *out) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (utt < 0 || utt >= D->q_n) return kamd::SetError(KAMD_ERR_ARG, "utterance %d outside the queue", utt);
  if (__atomic_load_n(&D->h_results[utt].status, __ATOMIC_ACQUIRE) != 1) return kamd::SetError(KAMD_ERR_STATE, "utterance %d has not finished", utt);
  *out = D->h_results[utt];
  return KAMD_OK;
}

int kamd_decoder_queue_fetch_lattice(kamd_decoder *h, int32_t utt, void *copy_stream, int32_t *num_states, int32_t *num_arcs,
                                     int32_t *start, int32_t **state_frame, int32_t **state_hclg, float **state_cost,
                                     float **state_final, kamd_lat_arc **arcs) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  kamd_queue_result r;
  int rc = kamd_decoder_queue_result(h, utt, &r);
  if (rc != KAMD_OK) return rc;
  *num_states = 0; *num_arcs = 0; *start = -1;
  *state_frame = NULL; *state_hclg = NULL; *state_cost = NULL; *state_final = NULL; *arcs = NULL;
  if (r.error)
    return kamd::SetError(KAMD_ERR_CAPACITY, "utterance %d: device capacity exceeded (flags %d:%s%s%s%s%s%s%s) at frame %d; raise kamd_decoder_sizes / the lattice pool",
                          utt, r.error, (r.error & 1) ? " hash" : "", (r.error & 2) ? " token-arena" : "", (r.error & 4) ? " link-arena" : "",
                          (r.error & 8) ? " max-frames" : "", (r.error & 16) ? " worklist" : "", (r.error & 32) ? " internal" : "",
                          (r.error & 64) ? " lattice-pool" : "", r.n_frames);
  const int F = r.n_frames, nt = r.n_tok, nl = r.n_link, n_last = r.n_last;
  (void)copy_stream;                     // nothing to copy: the blob is in host memory already (kamd_decoder_queue_configure)
  const size_t bytes = static_cast<size_t>(r.blob_bytes);
  const unsigned char *blob = D->h_pool + r.blob_off;
  if (bytes < (static_cast<size_t>(F + 2) + 2ull * nt + n_last + 6ull * nl) * 4) return kamd::SetError(KAMD_ERR_STATE, "utterance %d: short lattice blob", utt);
  const int *toff = reinterpret_cast<const int *>(blob);
  const int *st = toff + (F + 2);
  const float *co = reinterpret_cast<const float *>(st + nt);
  const float *lf = co + nt;
  const kamd::Link *lk = reinterpret_cast<const kamd::Link *>(lf + n_last);
  RawLat R;
  rc = Canonicalize(nt, nl, F, st, co, toff, lf, lk, 0, D->g->dev.start, &R);
  if (rc != KAMD_OK) return rc;
  const size_t n = R.frame.size(), m = R.arcs.size();
  *num_states = static_cast<int32_t>(n); *num_arcs = static_cast<int32_t>(m); *start = R.start;
  auto dup = [](const void *src, size_t bytes) -> void * { void *p = malloc(bytes ? bytes : 4); if (p && bytes) memcpy(p, src, bytes); return p; };
  *state_frame = static_cast<int32_t *>(dup(R.frame.data(), n * 4)); *state_hclg = static_cast<int32_t *>(dup(R.hclg.data(), n * 4));
  *state_cost = static_cast<float *>(dup(R.cost.data(), n * 4)); *state_final = static_cast<float *>(dup(R.fin.data(), n * 4));
  *arcs = static_cast<kamd_lat_arc *>(dup(R.arcs.data(), m * sizeof(kamd_lat_arc)));
  if (!*state_frame || !*state_hclg || !*state_cost || !*state_final || !*arcs) return kamd::SetError(KAMD_ERR_ARG, "out of host memory");
  return KAMD_OK;
}

int kamd_decoder_queue_wait(kamd_decoder *h, float *ms, int32_t *lanes) {
  Decoder *D = reinterpret_cast<Decoder *>(h);
  if (D->q_n <= 0) return kamd::SetError(KAMD_ERR_STATE, "no queue launched");
  KAMD_HIP(hipStreamSynchronize(D->q_stream));
  float t = 0;
  KAMD_HIP(hipEventElapsedTime(&t, D->qev[0], D->qev[1]));
  if (ms) *ms = t;
  if (lanes) *lanes = D->q_lanes;
  return KAMD_OK;
}

}  // extern "C"
