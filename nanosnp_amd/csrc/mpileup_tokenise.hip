// mpileup_tokenise.hip -- samtools-mpileup text -> (position, reference byte, column-5 string) per line, on the device.
//
// Replaces, for text resident in HBM, the reader in front of TensorMaker::make_tensor:
//   LineReader::getline    dna_sv_tensor/src/common/line_reader.cpp:95-127   lines end at '\n' (or at the end of the text); the
//                                                                            '\n' and ONE '\r' in front of it are dropped
//   split_line(.., "\t")   dna_sv_tensor/src/common/cpp_aux.cpp:43-59        tokens = maximal runs of non-tab bytes
//   create_pileup_tensor   make_candidate_snp_tensor/main.cpp:162-172        ref_off = atoll(token 1), ref base = chr_seq[ref_off-1],
//                                                                            pileup_bases = token 4
// and nsnp_mpileup_parse_into of the host library (nsnp_textio.c), which does the same on host cores: on a 6 M-column contig the host
// parse was the slowest station of the text path (17 ms beside 13 ms of device time) and got slower with every rank sharing the host.
// Here the raw text crosses PCIe as it is and the device cuts it.
//
// The grammar of a line is sequential only through TWO small pieces of state: how many tokens have started since the last newline
// (saturating at 7; token 1 = position, token 4 = bases) and how many newlines came before (the line's column index).  Both are
// prefix "sums" under an associative operator, so the text is cut into 8 KB tiles (256 threads x 32 bytes) and processed in five
// launches, every one a coalesced scan of the text or of an array an eighth of its size or smaller:
//   k_tok_summary   per tile: newlines, token-start state                      (reads the text once)
//   k_tok_scan      one workgroup: exclusive scan of the tile summaries
//   k_tok_lines     per tile: every byte's token index -> bit mask of the bytes inside a token 4 (one bit per byte, to a bitmap),
//                   bytes per tile; the thread that meets the start of a token 1 converts it (atoll) and writes pos / ref of its line;
//                   lines the reference could not read (fewer than five tokens, empty) raise status bits      (reads the text again)
//   k_tok_scan2     one workgroup: exclusive scan of the bytes per tile; totals and status -> meta
//   k_tok_compact   per tile: the token-4 bytes compacted through LDS into `bases` with 16-byte stores, col_off of the columns that
//                   start in the tile                                                       (reads the text a third time, + the bitmap)
// Algorithmic bytes per text byte: 3 reads + 1/8 bitmap write + 1/8 read, ~0.37 written as bases: ~3.6 B per text byte.  No line-length
// limit, no slow path: a line may span any number of tiles.  This five-launch form is the plain statement of the algorithm (option
// "tok_fused" 0); the default is k_tok_fused below - the same grammar as ONE launch that reads the text once.
#include "nsnp_common.hpp"

namespace {

constexpr int TK_BLOCK = 256;
constexpr int TK_CHUNK = 32;                       // bytes per thread: one 32-bit mask per byte class
constexpr int TK_TILE = TK_BLOCK * TK_CHUNK;       // 8 KB
constexpr int TK_SAT = 7;                          // token-start counts saturate here (only 0, 1..4, 2, 5 and "more" matter)

// the text as the kernels see it: base is 16-byte aligned, the text's bytes are base[lo, hi), hi > lo; everything outside reads as a
// tab, except one virtual newline at hi when the text does not end with one (line_reader.cpp:113: the end of the file ends a line)
struct TokText { const uint8_t* base; int64_t lo, hi; };

__device__ __forceinline__ int tk_byte(const TokText& t, int64_t p)
{
    if (p >= t.lo && p < t.hi) return t.base[p];
    return (p == t.hi && t.base[t.hi - 1] != '\n') ? '\n' : '\t';
}

// 4-bit mask of the bytes of w equal to the byte replicated in pat (exact: no borrow between the bytes)
__device__ __forceinline__ uint32_t eq4(uint32_t w, uint32_t pat)
{
    const uint32_t x = w ^ pat;
    uint32_t z = (x & 0x7f7f7f7fu) + 0x7f7f7f7fu;
    z = ~(z | x | 0x7f7f7f7fu);                     // 0x80 in every byte of x that is zero
    return (((z >> 7) * 0x00204081u) >> 21) & 0xfu;
}

struct TokMasks { uint32_t nl, sep, ts; };

// the 32 bytes at p0 (as words), the byte masks of the chunk: newlines, separators (tab, newline, a '\r' right in front of a
// newline), token starts (a non-separator behind a separator)
__device__ __forceinline__ TokMasks tk_load(const TokText& t, int64_t p0, uint32_t (&w)[8])
{
    if (p0 >= t.lo && p0 + TK_CHUNK <= t.hi) {
        const uint4 a = *reinterpret_cast<const uint4*>(t.base + p0), b = *reinterpret_cast<const uint4*>(t.base + p0 + 16);
        w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
    } else if (p0 + TK_CHUNK <= t.lo || p0 > t.hi) {
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = 0x09090909u;
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t v = 0;
            for (int j = 0; j < 4; ++j) v |= (uint32_t)tk_byte(t, p0 + 4 * k + j) << (8 * j);
            w[k] = v;
        }
    }
    uint32_t nl = 0, tab = 0, cr = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        nl |= eq4(w[k], 0x0a0a0a0au) << (4 * k);
        tab |= eq4(w[k], 0x09090909u) << (4 * k);
        cr |= eq4(w[k], 0x0d0d0d0du) << (4 * k);
    }
    const int nextb = tk_byte(t, p0 + TK_CHUNK), prevb = tk_byte(t, p0 - 1);
    const uint32_t nl_next = (nl >> 1) | (nextb == '\n' ? 0x80000000u : 0u);
    TokMasks m;
    m.nl = nl;
    m.sep = tab | nl | (cr & nl_next);
    const bool prev_sep = prevb == '\t' || prevb == '\n' || (prevb == '\r' && (nl & 1u));
    m.ts = ~m.sep & ((m.sep << 1) | (prev_sep ? 1u : 0u));
    return m;
}

// token-start state: bit 3 = a newline was seen, bits 0-2 = token starts since the last newline (or since the beginning), saturating
__device__ __forceinline__ int st_combine(int a, int b)
{
    if (b & 8) return b;
    const int c = (a & 7) + (b & 7);
    return (a & 8) | (c > TK_SAT ? TK_SAT : c);
}
__device__ __forceinline__ int st_of(const TokMasks& m)
{
    if (!m.nl) { const int c = __popc(m.ts); return c > TK_SAT ? TK_SAT : c; }
    const int top = 31 - __clz(m.nl);
    const uint32_t behind = top == 31 ? 0u : (m.ts & (0xffffffffu << (top + 1)));
    const int c = __popc(behind);
    return 8 | (c > TK_SAT ? TK_SAT : c);
}

// exclusive scans over the 256 threads of a workgroup: a sum (v) and the state operator (st); totals of both
struct BlockScan { int v_excl, st_excl, v_total, st_total; };
__device__ __forceinline__ BlockScan tk_block_scan(int v, int st, int (*sh)[2])          // sh[TK_BLOCK / 64][2]
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int vi = v, si = st;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int pv = __shfl_up(vi, o), ps = __shfl_up(si, o);
        if (lane >= o) { vi += pv; si = st_combine(ps, si); }
    }
    if (lane == 63) { sh[wave][0] = vi; sh[wave][1] = si; }
    __syncthreads();
    int vb = 0, sb = 0;
    for (int k = 0; k < wave; ++k) { vb += sh[k][0]; sb = st_combine(sb, sh[k][1]); }
    BlockScan r;
    int ve = __shfl_up(vi, 1), se = __shfl_up(si, 1);
    if (lane == 0) { ve = 0; se = 0; }
    r.v_excl = vb + ve;
    r.st_excl = st_combine(sb, se);
    int vt = 0, stt = 0;
    for (int k = 0; k < TK_BLOCK / 64; ++k) { vt += sh[k][0]; stt = st_combine(stt, sh[k][1]); }
    r.v_total = vt; r.st_total = stt;
    __syncthreads();                               // (sh may be reused by the caller's next scan)
    return r;
}

__global__ __launch_bounds__(TK_BLOCK) void k_tok_summary(TokText t, int64_t* __restrict__ tile_nl, int32_t* __restrict__ tile_st)
{
    __shared__ int sh[TK_BLOCK / 64][2];
    const int64_t p0 = (int64_t)blockIdx.x * TK_TILE + threadIdx.x * TK_CHUNK;
    uint32_t w[8];
    const TokMasks m = tk_load(t, p0, w);
    const BlockScan s = tk_block_scan(__popc(m.nl), st_of(m), sh);
    if (threadIdx.x == 0) { tile_nl[blockIdx.x] = s.v_total; tile_st[blockIdx.x] = s.st_total; }
}

// one workgroup: tile_nl -> newlines in front of every tile (in place), tile_st -> token starts since the last newline in front of
// every tile (in place); the number of lines -> ws_meta[0]; the status word ws_meta[2] starts at zero
__global__ __launch_bounds__(1024) void k_tok_scan(int64_t* __restrict__ tile_nl, int32_t* __restrict__ tile_st, int64_t n_tiles,
                                                    int64_t* __restrict__ ws_meta)
{
    __shared__ int64_t part[1024];
    __shared__ int pst[1024];
    const int tid = threadIdx.x;
    const int64_t per = NSNP_CDIV(n_tiles, 1024);
    const int64_t b0 = tid * per, b1 = (b0 + per < n_tiles) ? b0 + per : n_tiles;
    int64_t s = 0; int st = 0;
    for (int64_t b = b0; b < b1; ++b) { s += tile_nl[b]; st = st_combine(st, tile_st[b]); }
    part[tid] = s; pst[tid] = st;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t v = tid >= o ? part[tid - o] : 0;
        const int q = tid >= o ? pst[tid - o] : 0;
        __syncthreads();
        part[tid] += v; pst[tid] = st_combine(q, pst[tid]);
        __syncthreads();
    }
    int64_t run = tid ? part[tid - 1] : 0;
    int rst = tid ? pst[tid - 1] : 0;
    for (int64_t b = b0; b < b1; ++b) {
        const int64_t v = tile_nl[b]; const int q = tile_st[b];
        tile_nl[b] = run; tile_st[b] = rst & 7;
        run += v; rst = st_combine(rst, q);
    }
    if (tid == 1023) { ws_meta[0] = part[1023]; ws_meta[2] = 0; }
}

enum { TOK_EFORMAT = NSNP_TOK_EFORMAT, TOK_BLANK = NSNP_TOK_BLANK, TOK_EPOS = NSNP_TOK_EPOS, TOK_ERANGE = NSNP_TOK_ERANGE };

// atoll on the token that starts at p (main.cpp:165): white space, one sign, digits; never beyond the token's end.  The bytes of the
// tile itself come from its copy in LDS (txt: the 8 KB the workgroup loaded, virtual bytes included) - a lane walks its digits one
// dependent read at a time, and from global memory that walk was most of the kernel's time (74 of 180 us per 64 MB) -; a token that
// runs beyond the tile continues in global memory.
__device__ __forceinline__ int64_t tk_atoll(const TokText& t, const uint8_t* txt, int64_t tile0, int64_t p)
{
    auto rd = [&](int64_t q) -> int { const int64_t o = q - tile0; return (o >= 0 && o < TK_TILE) ? (int)txt[o] : tk_byte(t, q); };
    auto ends = [&](int64_t q, int ch) { return ch == '\t' || ch == '\n' || (ch == '\r' && rd(q + 1) == '\n'); };
    int ch = rd(p);
    while (!ends(p, ch) && (ch == ' ' || ch == '\r' || ch == '\v' || ch == '\f')) ch = rd(++p);
    bool neg = false;
    if (!ends(p, ch) && (ch == '-' || ch == '+')) { neg = ch == '-'; ch = rd(++p); }
    uint64_t v = 0;
    while (ch >= '0' && ch <= '9') { v = v * 10u + (uint64_t)(ch - '0'); ch = rd(++p); }
    return neg ? (int64_t)(0ull - v) : (int64_t)v;
}
__device__ __forceinline__ void tk_stage_text(uint8_t* txt, const uint32_t (&w)[8])
{
    uint4* d = reinterpret_cast<uint4*>(txt + threadIdx.x * TK_CHUNK);
    d[0] = uint4{w[0], w[1], w[2], w[3]}; d[1] = uint4{w[4], w[5], w[6], w[7]};
}

__global__ __launch_bounds__(TK_BLOCK) void k_tok_lines(TokText t, const int64_t* __restrict__ tile_nl, const int32_t* __restrict__ tile_st,
                                                         const uint8_t* __restrict__ chr_seq, int64_t chr_len, int64_t cap_cols,
                                                         int64_t* __restrict__ pos, uint8_t* __restrict__ ref,
                                                         uint32_t* __restrict__ bitmap, int64_t* __restrict__ tile_bytes,
                                                         int64_t* __restrict__ ws_meta)
{
    __shared__ int sh[TK_BLOCK / 64][2];
    __shared__ __attribute__((aligned(16))) uint8_t txt[TK_TILE];
    const int64_t p0 = (int64_t)blockIdx.x * TK_TILE + threadIdx.x * TK_CHUNK;
    uint32_t w[8];
    const TokMasks m = tk_load(t, p0, w);
    tk_stage_text(txt, w);
    const BlockScan s = tk_block_scan(__popc(m.nl), st_of(m), sh);      // (its barriers publish txt)
    int cc = st_combine(tile_st[blockIdx.x], s.st_excl) & 7;       // token starts since the last newline in front of this chunk
    const int64_t line0 = tile_nl[blockIdx.x] + s.v_excl;          // newlines in front of this chunk = index of the line it starts in
    uint32_t ev = m.ts | m.nl, m4 = 0, err = 0;
    int prevb = 0;
    while (ev) {
        const int b = __ffs(ev) - 1;
        ev &= ev - 1;
        if (cc == 5) m4 |= (b ? (0xffffffffu >> (32 - b)) : 0u) & (0xffffffffu << prevb);      // bytes [prevb, b) belong to a token 4
        if ((m.nl >> b) & 1u) {
            if (cc == 0) err |= TOK_BLANK; else if (cc < 5) err |= TOK_EFORMAT;
            cc = 0;
        } else {
            cc = cc < TK_SAT ? cc + 1 : TK_SAT;
            if (cc == 2) {
                const int64_t line = line0 + __popc(m.nl & ((1u << b) - 1u));
                const int64_t v = tk_atoll(t, txt, (int64_t)blockIdx.x * TK_TILE, p0 + b);
                if (line < cap_cols) {
                    pos[line] = v;
                    if (ref) {
                        if (v >= 1 && v <= chr_len) ref[line] = chr_seq[v - 1];
                        else { ref[line] = 'N'; err |= TOK_EPOS; }
                    }
                }
            }
        }
        prevb = b;
    }
    if (cc == 5) m4 |= 0xffffffffu << prevb;
    m4 &= ~m.sep;
    bitmap[(int64_t)blockIdx.x * TK_BLOCK + threadIdx.x] = m4;
    if (err) atomicOr(reinterpret_cast<unsigned long long*>(ws_meta + 2), (unsigned long long)err);
    // bytes of this tile that belong to a token 4
    int n = __popc(m4);
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][0] = n;
    __syncthreads();
    if (threadIdx.x == 0) { int64_t tot = 0; for (int k = 0; k < TK_BLOCK / 64; ++k) tot += sh[k][0]; tile_bytes[blockIdx.x] = tot; }
}

// one workgroup: tile_bytes -> token-4 bytes in front of every tile (in place); totals and status -> meta (any device-visible memory)
__global__ __launch_bounds__(1024) void k_tok_scan2(int64_t* __restrict__ tile_bytes, int64_t n_tiles, int64_t* __restrict__ ws_meta,
                                                     int64_t cap_cols, int64_t cap_bytes, int64_t* __restrict__ col_off,
                                                     int64_t* __restrict__ meta)
{
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int64_t per = NSNP_CDIV(n_tiles, 1024);
    const int64_t b0 = tid * per, b1 = (b0 + per < n_tiles) ? b0 + per : n_tiles;
    int64_t s = 0;
    for (int64_t b = b0; b < b1; ++b) s += tile_bytes[b];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int64_t v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int64_t run = tid ? part[tid - 1] : 0;
    for (int64_t b = b0; b < b1; ++b) { const int64_t v = tile_bytes[b]; tile_bytes[b] = run; run += v; }
    if (tid == 1023) {
        const int64_t n_cols = ws_meta[0], n_bytes = part[1023];
        int64_t status = ws_meta[2];
        if (n_cols > cap_cols || n_bytes > cap_bytes) status |= TOK_ERANGE;
        else col_off[n_cols] = n_bytes;
        ws_meta[1] = n_bytes;
        meta[0] = n_cols; meta[1] = n_bytes; meta[2] = status; meta[3] = 0;
    }
}

__global__ __launch_bounds__(TK_BLOCK) void k_tok_compact(TokText t, const int64_t* __restrict__ tile_nl, const int64_t* __restrict__ tile_bytes,
                                                           const uint32_t* __restrict__ bitmap, int64_t cap_cols, int64_t cap_bytes,
                                                           int64_t* __restrict__ col_off, uint8_t* __restrict__ bases)
{
    __shared__ int sh[TK_BLOCK / 64][2];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[TK_TILE + 32];
    const int64_t idx = (int64_t)blockIdx.x * TK_BLOCK + threadIdx.x;
    const int64_t p0 = idx * TK_CHUNK;
    uint32_t w[8];
    const TokMasks m = tk_load(t, p0, w);
    const uint32_t m4 = bitmap[idx];
    const uint32_t prev_in = idx ? (bitmap[idx - 1] >> 31) : 0u;
    // two sums at once: newlines and token-4 bytes in front of this chunk (the state slot of the scan is unused: newline-free states add)
    const int n4 = __popc(m4);
    int vi = (__popc(m.nl) << 16) | n4;            // <= 32 each per thread, <= 8192 per tile: 16-bit fields suffice
    const BlockScan s = tk_block_scan(vi, 0, sh);
    const int nl_excl = s.v_excl >> 16, r0 = s.v_excl & 0xffff, tile_cnt = s.v_total & 0xffff;
    const int64_t out0 = tile_bytes[blockIdx.x];
    const int mis = (int)(out0 & 15);
    // columns that start in this chunk
    uint32_t starts = m4 & ~((m4 << 1) | prev_in);
    while (starts) {
        const int b = __ffs(starts) - 1;
        starts &= starts - 1;
        const uint32_t below = (1u << b) - 1u;
        const int64_t line = tile_nl[blockIdx.x] + nl_excl + __popc(m.nl & below);
        if (line < cap_cols) col_off[line] = out0 + r0 + __popc(m4 & below);
    }
    // the token-4 bytes of the tile, compacted: cbuf[mis + rank]
    {
        uint32_t rest = m4; int r = mis + r0;
        while (rest) {
            const int b = __ffs(rest) - 1;
            rest &= rest - 1;
            cbuf[r++] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
    __syncthreads();
    // 16-byte pieces of bases[out0 - mis, out0 + tile_cnt): whole pieces as one store, the two ragged ones byte by byte
    const int span = mis + tile_cnt;
    uint8_t* __restrict__ gb = bases + (out0 - mis);
    for (int o = threadIdx.x * 16; o < span; o += TK_BLOCK * 16) {
        const bool whole = o >= mis && o + 16 <= span && out0 - mis + o + 16 <= cap_bytes && ((uintptr_t)(gb + o) & 15) == 0;
        if (whole) *reinterpret_cast<uint4*>(gb + o) = *reinterpret_cast<const uint4*>(cbuf + o);
        else {
            for (int k = 0; k < 16; ++k) {
                const int q = o + k;
                if (q >= mis && q < span && out0 - mis + q < cap_bytes) gb[q] = cbuf[q];
            }
        }
    }
}


// ---- the same in ONE launch: a chained scan (decoupled look-back) -------------------------------------------------------------------
// The five launches above read the text three times and spend two launches on single-workgroup scans: 180 us per 64 MB chunk, of which
// the text itself is 25 us at HBM speed.  Here a tile keeps its 32 bytes per thread in registers, publishes its (newlines, token-start
// state) aggregate, looks back over its predecessors' aggregates / prefixes for the line index and the token-start count at its first
// byte, marks column 5, publishes its byte count, looks back for its output offset and compacts - the text is read once.
// Descriptors are single 64-bit words:
//   a: [63:62] 0 empty / 1 aggregate / 2 inclusive prefix, [61:58] token-start state, [57:0] newlines;
//   b: [63:62] likewise, [61:58] status bits met so far (OR), [57:0] column-5 bytes
// Tiles are taken in blockIdx order (workgroups are dispatched in that order, so a tile's predecessors have always started); the last
// tile's inclusive prefixes are the totals and the status of the whole text: no counter, no atomic read-modify-write anywhere.
struct TokDesc { unsigned long long a, b; };
constexpr unsigned long long TD_AGG = 1ull << 62, TD_PFX = 2ull << 62;
__device__ __forceinline__ unsigned long long td_load(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void td_store(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// one wave: exclusive prefix of tile `tile` over descriptor word `word` (0: a = state + newlines, 1: b = bytes); every lane returns it
__device__ __forceinline__ void td_lookback(const TokDesc* desc, long long tile, int word, long long& sum_out, int& st_out)
{
    const int lane = threadIdx.x & 63;
    long long run = 0; int run_st = 0;
    for (long long base = tile - 1; base >= 0; base -= 64) {
        const long long j = base - lane;
        unsigned long long d;
        for (;;) {
            d = j >= 0 ? td_load(word ? &desc[j].b : &desc[j].a) : TD_PFX;     // (in front of tile 0: an empty prefix)
            if (!__any((d >> 62) == 0)) break;
            __builtin_amdgcn_s_sleep(1);
        }
        const unsigned long long pm = __ballot((d >> 62) == 2);
        const int P = pm ? __builtin_ctzll(pm) : 64;                            // the nearest tile that knows its inclusive prefix
        long long v = lane <= P ? (long long)(d & ((1ull << 58) - 1)) : 0;
        int st = lane <= P ? (int)((d >> 58) & 15) : 0;          // word 0: token-start state (st_combine); word 1: status bits (OR)
        // lane l holds tile base - l: the window's value is  d[P] o ... o d[1] o d[0]  (earlier tiles on the left)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long ov = __shfl_down(v, o); const int os = __shfl_down(st, o);
            if (lane + o < 64) { v += ov; st = word ? (os | st) : st_combine(os, st); }
        }
        v = __shfl(v, 0); st = __shfl(st, 0);
        run += v; run_st = word ? (st | run_st) : st_combine(st, run_st);
        if (pm) break;
    }
    sum_out = run; st_out = run_st;
}

__global__ __launch_bounds__(TK_BLOCK) void k_tok_fused(TokText t, long long n_tiles, TokDesc* __restrict__ desc,
                                                         const uint8_t* __restrict__ chr_seq, int64_t chr_len, int64_t cap_cols, int64_t cap_bytes,
                                                         int64_t* __restrict__ pos, uint8_t* __restrict__ ref, int64_t* __restrict__ col_off,
                                                         uint8_t* __restrict__ bases, int64_t* __restrict__ meta)
{
    __shared__ int sh[TK_BLOCK / 64][2];
    __shared__ long long sh_b[4];
    __shared__ int sh_err[TK_BLOCK / 64];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[TK_TILE + 32];
    const int tid = threadIdx.x, wave = tid >> 6;
    const long long tile = blockIdx.x;
    const int64_t p0 = tile * TK_TILE + tid * TK_CHUNK;
    uint32_t w[8];
    const TokMasks m = tk_load(t, p0, w);
    tk_stage_text(cbuf, w);                          // the tile's bytes for tk_atoll (cbuf is free until the compaction: two barriers further down)
    const BlockScan s = tk_block_scan(__popc(m.nl), st_of(m), sh);
    // ---- newlines and token-start state in front of this tile ----
    if (wave == 0) {
        if (tid == 0) td_store(&desc[tile].a, (tile == 0 ? TD_PFX : TD_AGG) | ((unsigned long long)s.st_total << 58) | (unsigned long long)s.v_total);
        long long nl0 = 0; int st0 = 0;
        if (tile > 0) {
            td_lookback(desc, tile, 0, nl0, st0);
            if (tid == 0) td_store(&desc[tile].a, TD_PFX | ((unsigned long long)st_combine(st0, s.st_total) << 58) | (unsigned long long)(nl0 + s.v_total));
        }
        if (tid == 0) { sh_b[1] = nl0; sh_b[2] = st0; }
    }
    __syncthreads();
    const int64_t line0 = sh_b[1] + s.v_excl;
    int cc = st_combine((int)sh_b[2], s.st_excl) & 7;
    // ---- the grammar over this chunk's events: column-5 bytes, their starts, the position of every line that starts its token 1 here ----
    uint32_t ev = m.ts | m.nl, m4 = 0, s4 = 0, err = 0;
    int prevb = 0;
    while (ev) {
        const int b = __ffs(ev) - 1;
        ev &= ev - 1;
        if (cc == 5) m4 |= (b ? (0xffffffffu >> (32 - b)) : 0u) & (0xffffffffu << prevb);
        if ((m.nl >> b) & 1u) {
            if (cc == 0) err |= TOK_BLANK; else if (cc < 5) err |= TOK_EFORMAT;
            cc = 0;
        } else {
            cc = cc < TK_SAT ? cc + 1 : TK_SAT;
            if (cc == 5) s4 |= 1u << b;
            if (cc == 2) {
                const int64_t line = line0 + __popc(m.nl & ((1u << b) - 1u));
                const int64_t v = tk_atoll(t, cbuf, tile * TK_TILE, p0 + b);
                if (line < cap_cols) {
                    pos[line] = v;
                    if (ref) {
                        if (v >= 1 && v <= chr_len) ref[line] = chr_seq[v - 1];
                        else { ref[line] = 'N'; err |= TOK_EPOS; }
                    }
                }
            }
        }
        prevb = b;
    }
    if (cc == 5) m4 |= 0xffffffffu << prevb;
    m4 &= ~m.sep;
    // ---- column-5 bytes in front of this tile; the status bits of the tile ride along ----
    {
        int e = (int)err;
        for (int o = 32; o > 0; o >>= 1) e |= __shfl_xor(e, o);
        if ((tid & 63) == 0) sh_err[wave] = e;
    }
    const BlockScan sb = tk_block_scan(__popc(m4), 0, sh);             // (its barriers publish sh_err)
    const int r0 = sb.v_excl, tile_cnt = sb.v_total;
    if (wave == 0) {
        int terr = 0;
        for (int k = 0; k < TK_BLOCK / 64; ++k) terr |= sh_err[k];
        if (tid == 0) td_store(&desc[tile].b, (tile == 0 ? TD_PFX : TD_AGG) | ((unsigned long long)terr << 58) | (unsigned long long)tile_cnt);
        long long b0 = 0; int perr = 0;
        if (tile > 0) {
            td_lookback(desc, tile, 1, b0, perr);
            if (tid == 0) td_store(&desc[tile].b, TD_PFX | ((unsigned long long)(perr | terr) << 58) | (unsigned long long)(b0 + tile_cnt));
        }
        if (tid == 0) { sh_b[3] = b0; sh_b[0] = perr | terr; }
    }
    __syncthreads();
    const int64_t out0 = sh_b[3];
    const int mis = (int)(out0 & 15);
    while (s4) {
        const int b = __ffs(s4) - 1;
        s4 &= s4 - 1;
        const uint32_t below = (1u << b) - 1u;
        const int64_t line = line0 + __popc(m.nl & below);
        if (line < cap_cols) col_off[line] = out0 + r0 + __popc(m4 & below);
    }
    {
        uint32_t rest = m4; int r = mis + r0;
        while (rest) {
            const int b = __ffs(rest) - 1;
            rest &= rest - 1;
            cbuf[r++] = (uint8_t)(w[b >> 2] >> (8 * (b & 3)));
        }
    }
    __syncthreads();
    const int span = mis + tile_cnt;
    uint8_t* __restrict__ gb = bases + (out0 - mis);
    for (int o = tid * 16; o < span; o += TK_BLOCK * 16) {
        const bool whole = o >= mis && o + 16 <= span && out0 - mis + o + 16 <= cap_bytes && ((uintptr_t)(gb + o) & 15) == 0;
        if (whole) *reinterpret_cast<uint4*>(gb + o) = *reinterpret_cast<const uint4*>(cbuf + o);
        else {
            for (int k = 0; k < 16; ++k) {
                const int q = o + k;
                if (q >= mis && q < span && out0 - mis + q < cap_bytes) gb[q] = cbuf[q];
            }
        }
    }
    // ---- the last tile's inclusive prefixes are the totals and the status of the whole text ----
    if (tid == 0 && tile == n_tiles - 1) {
        const int64_t n_cols = sh_b[1] + s.v_total, n_bytes = out0 + tile_cnt;
        int64_t status = sh_b[0];
        if (n_cols > cap_cols || n_bytes > cap_bytes) status |= TOK_ERANGE;
        else col_off[n_cols] = n_bytes;
        meta[0] = n_cols; meta[1] = n_bytes; meta[2] = status; meta[3] = 0;
    }
}

}  // namespace

// workspace of the tokeniser: grows when a longer text than ever before arrives (synchronous, like the selection scratch)
static int tok_reserve(nsnp_ctx* ctx, int64_t n_tiles, hipStream_t s)
{
    const size_t need = (size_t)n_tiles * (8 + 8 + 4 + 4 * TK_BLOCK) + 64 + 256;       // (the one-launch form needs 16 bytes per tile + 64: less)
    if (ctx->tok_ws_bytes >= need) return NSNP_OK;
    NSNP_HIP(ctx, hipStreamSynchronize(s));
    if (ctx->tok_ws) (void)hipFree(ctx->tok_ws);
    ctx->tok_ws = nullptr; ctx->tok_ws_bytes = 0;
    const size_t want = need + need / 4;
    NSNP_HIP(ctx, hipMalloc(&ctx->tok_ws, want));
    ctx->tok_ws_bytes = want;
    return NSNP_OK;
}

void nsnp_tok_free(nsnp_ctx* ctx)
{
    if (ctx->tok_ws) (void)hipFree(ctx->tok_ws);
    ctx->tok_ws = nullptr; ctx->tok_ws_bytes = 0;
}

extern "C" int nsnp_mpileup_tokenise(nsnp_ctx* ctx, const uint8_t* text, int64_t text_len, const uint8_t* chr_seq, int64_t chr_len,
                                     int64_t cap_cols, int64_t cap_bytes, int64_t* pos, int64_t* col_off, uint8_t* bases, uint8_t* ref,
                                     int64_t* meta, void* stream)
{
    if (!ctx || text_len < 0 || cap_cols < 0 || cap_bytes < 0 || !meta || !col_off || (text_len > 0 && !text) ||
        (cap_cols > 0 && !pos) || (cap_bytes > 0 && !bases) || (ref && (!chr_seq || chr_len < 0)))
        return NSNP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int mis = (int)((uintptr_t)text & 15);
    TokText t;
    t.base = text - mis; t.lo = mis; t.hi = mis + text_len;
    const int64_t n_tiles = NSNP_CDIV(t.hi + 1, (int64_t)TK_TILE);
    int rc = tok_reserve(ctx, n_tiles, s);
    if (rc != NSNP_OK) return rc;
    uint8_t* ws = (uint8_t*)ctx->tok_ws;
    int64_t* ws_meta = (int64_t*)ws;                               // [0] lines, [1] bytes, [2] status
    int64_t* tile_nl = (int64_t*)(ws + 64);
    int64_t* tile_bytes = tile_nl + n_tiles;
    uint32_t* bitmap = (uint32_t*)(tile_bytes + n_tiles);
    int32_t* tile_st = (int32_t*)(bitmap + n_tiles * TK_BLOCK);
    if (text_len == 0) {
        NSNP_HIP(ctx, hipMemsetAsync(meta, 0, 4 * sizeof(int64_t), s));
        NSNP_HIP(ctx, hipMemsetAsync(col_off, 0, sizeof(int64_t), s));
        return NSNP_OK;
    }
    if (ctx->tok_fused) {
        // one launch: its descriptors (16 bytes per tile) zeroed in front of it
        TokDesc* desc = (TokDesc*)(ws + 64);
        NSNP_HIP(ctx, hipMemsetAsync(desc, 0, (size_t)n_tiles * sizeof(TokDesc), s));
        hipLaunchKernelGGL(k_tok_fused, dim3((unsigned)n_tiles), dim3(TK_BLOCK), 0, s, t, (long long)n_tiles, desc, chr_seq, chr_len, cap_cols, cap_bytes,
                           pos, ref, col_off, bases, meta);
        NSNP_HIP(ctx, hipGetLastError());
        return NSNP_OK;
    }
    hipLaunchKernelGGL(k_tok_summary, dim3((unsigned)n_tiles), dim3(TK_BLOCK), 0, s, t, tile_nl, tile_st);
    hipLaunchKernelGGL(k_tok_scan, dim3(1), dim3(1024), 0, s, tile_nl, tile_st, n_tiles, ws_meta);
    hipLaunchKernelGGL(k_tok_lines, dim3((unsigned)n_tiles), dim3(TK_BLOCK), 0, s, t, (const int64_t*)tile_nl, (const int32_t*)tile_st,
                       chr_seq, chr_len, cap_cols, pos, ref, bitmap, tile_bytes, ws_meta);
    hipLaunchKernelGGL(k_tok_scan2, dim3(1), dim3(1024), 0, s, tile_bytes, n_tiles, ws_meta, cap_cols, cap_bytes, col_off, meta);
    hipLaunchKernelGGL(k_tok_compact, dim3((unsigned)n_tiles), dim3(TK_BLOCK), 0, s, t, (const int64_t*)tile_nl, (const int64_t*)tile_bytes,
                       (const uint32_t*)bitmap, cap_cols, cap_bytes, col_off, bases);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
