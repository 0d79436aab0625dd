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
// (saturating at 7; token 1 = position, token 4 = bases) and how many newlines came before (the line's column index).  The first is
// LOCAL: a tile (8 KB: 256 threads x 32 bytes) finds it in the text just in front of itself (the 2 KB before it hold a newline unless a
// line is longer than that; then it looks further back).  The second and the output offset are prefix sums over the tiles.  Three
// launches, every one a coalesced scan, no workgroup ever waits for another:
//   k_tok_count   per tile: byte-class masks (SWAR compares on the 32 bytes a thread loaded), the token index of every byte through a
//                 workgroup scan of the token-start state, the bytes inside a token 4; -> newlines and column-5 bytes of the tile;
//                 lines the reference could not read (fewer than five tokens, empty) raise status bits          (reads the text once)
//   k_tok_scan    one workgroup: exclusive scans of both counts over the tiles; totals and status -> meta
//                 the four bit masks of every 32-byte chunk (newlines, column-5 bytes, starts of tokens 1 and 4) are kept for launch 3
//   k_tok_emit    per tile again, now knowing the line index and output offset of its first byte, with the chunk masks of launch 1: the
//                 thread that holds the start of a token 1 converts it (atoll, digits read from the tile's copy in LDS) and writes pos /
//                 ref of its line, col_off of the tokens 4 that start here, their bytes compacted through LDS into `bases` with
//                 16-byte stores                                                       (reads the text again + 16 B per 32 B of masks)
// (+ a one-thread launch that folds "position outside the reference", which only launch 3 can see, into meta.)  Algorithmic bytes: the
// text once + what is written (column-5 bytes, 17 bytes per line); the kernels read the text twice (+ 2 KB per tile for the local state).
// No line-length limit, no slow path.  k_tok_fused below is the same grammar as ONE launch (a chained scan; option "tok_fused" 1): with
// the chip to itself it is no faster than this form, and its tiles spin on their predecessors' descriptors - when four processes shared
// one GPU (tools/host_scaling.sh) a contig's tokenising took 40 s instead of 1.4 ms.  A kernel whose progress depends on how the device
// is shared is not a default.
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
__device__ __forceinline__ void tk_load_words(const TokText& t, int64_t p0, uint32_t (&w)[8])
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
}
__device__ __forceinline__ TokMasks tk_load(const TokText& t, int64_t p0, uint32_t (&w)[8])
{
    tk_load_words(t, p0, w);
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

// exclusive prefix of one value per thread over a workgroup of 1024 threads (16 waves): wave shuffles, one exchange through LDS.
// v: a sum; st: the token-start state operator (pass 0 where it is not needed).  Returns the exclusive prefixes and the totals.
struct WgScan { long long v_excl, v_total; int st_excl, st_total; };
__device__ __forceinline__ WgScan tk_wg_scan1024(long long v, int st, long long* sh_v, int* sh_s)      // sh_v[16], sh_s[16]
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long vi = v; int si = st;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long pv = __shfl_up(vi, o); const int ps = __shfl_up(si, o);
        if (lane >= o) { vi += pv; si = st_combine(ps, si); }
    }
    if (lane == 63) { sh_v[wave] = vi; sh_s[wave] = si; }
    __syncthreads();
    long long vb = 0, vt = 0; int sb = 0, stt = 0;
    for (int k = 0; k < 16; ++k) {
        if (k == wave) { vb = vt; sb = stt; }
        vt += sh_v[k]; stt = st_combine(stt, sh_s[k]);
    }
    long long ve = __shfl_up(vi, 1); int se = __shfl_up(si, 1);
    if (lane == 0) { ve = 0; se = 0; }
    WgScan r;
    r.v_excl = vb + ve; r.st_excl = st_combine(sb, se); r.v_total = vt; r.st_total = stt;
    __syncthreads();
    return r;
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


// token starts since the last newline in front of tile0, found in the text itself: one byte per lane, 64 bytes per window, nearest window
// first - two ballots give the newline and separator masks of a window as scalars, the token starts behind the last newline are a
// popcount; a window without a newline (a line longer than it) adds its starts and the walk goes on.  One whole wave calls this;
// every lane returns the count.  (The first form ran tk_load's full 32-bytes-per-lane mask arithmetic over the 2 KB in front of the
// tile: 300 vector instructions for one wave of every tile; this one is ~25 per window.)
__device__ __forceinline__ int tk_local_carry(const TokText& t, int64_t tile0)
{
    const int lane = threadIdx.x & 63;
    int run = 0;
    bool pending = false;                                  // the nearest byte seen so far is a non-separator: it starts a token iff the byte in front of it is one
    int next = tk_byte(t, tile0);                          // the byte behind the window
    for (int64_t end = tile0; end > t.lo; end -= 64) {
        const int ch = tk_byte(t, end - 64 + lane);
        int nx = __shfl_down(ch, 1);
        if (lane == 63) nx = next;
        const bool nl = ch == '\n';
        const bool sep = nl || ch == '\t' || (ch == '\r' && nx == '\n');
        const unsigned long long nlm = __ballot(nl), sepm = __ballot(sep);
        if (pending && (sepm >> 63)) ++run;                // the window's last byte is a separator: the pending byte behind it starts a token
        const unsigned long long ts = ~sepm & (sepm << 1);                                   // starts whose predecessor lies in this window
        if (nlm) {
            const int top = 63 - __builtin_clzll(nlm);
            run += top == 63 ? 0 : __popcll(ts >> (top + 1));
            return run > TK_SAT ? TK_SAT : run;
        }
        run += __popcll(ts);
        pending = !(sepm & 1ull);                          // lane 0's byte is a non-separator whose predecessor is in the next window
        if (run > TK_SAT) run = TK_SAT;
        next = __shfl(ch, 0);
    }
    if (pending) ++run;                                    // in front of the text: a separator (the text starts a line)
    return run > TK_SAT ? TK_SAT : run;
}

// the grammar over one chunk's events, given the token starts since the last newline in front of it: bit masks of the bytes inside a
// token 4 (m4), of the starts of tokens 4 (s4) and 1 (s2), status bits of the lines that end here
struct TokEvents { uint32_t m4, s4, s2, err; };
__device__ __forceinline__ TokEvents tk_events(const TokMasks& m, int cc)
{
    uint32_t ev = m.ts | m.nl, m4 = 0, s4 = 0, s2 = 0, err = 0;
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
            if (cc == 5) s4 |= 1u << b;
            if (cc == 2) s2 |= 1u << b;
        }
        prevb = b;
    }
    if (cc == 5) m4 |= 0xffffffffu << prevb;
    return TokEvents{m4 & ~m.sep, s4, s2, err};
}

// launch 1 of 3: per tile the newlines and the column-5 bytes it holds (and the token-start count at its first byte, kept for launch 3)
__global__ __launch_bounds__(TK_BLOCK) void k_tok_count(TokText t, int64_t* __restrict__ tile_nl, int64_t* __restrict__ tile_bytes,
                                                         uint4* __restrict__ chunk_masks, int64_t* __restrict__ ws_meta)
{
    __shared__ int sh[TK_BLOCK / 64][2];
    __shared__ int sh_carry;
    const int64_t tile0 = (int64_t)blockIdx.x * TK_TILE;
    uint32_t w[8];
    const TokMasks m = tk_load(t, tile0 + threadIdx.x * TK_CHUNK, w);
    if (threadIdx.x < 64) { const int c = blockIdx.x ? tk_local_carry(t, tile0) : 0; if (threadIdx.x == 0) sh_carry = c; }
    const BlockScan s = tk_block_scan(__popc(m.nl), st_of(m), sh);                      // (its barriers publish sh_carry)
    const TokEvents e = tk_events(m, st_combine(sh_carry, s.st_excl) & 7);
    if (e.err) atomicOr(reinterpret_cast<unsigned long long*>(ws_meta + 2), (unsigned long long)e.err);
    // what launch 3 needs of this chunk's grammar: 16 bytes per 32 bytes of text (it then skips the mask arithmetic, the state scan and
    // the event walk: 500 of its 1,100 vector instructions per wave; both kernels sit at their vector-issue bound)
    chunk_masks[(int64_t)blockIdx.x * TK_BLOCK + threadIdx.x] = uint4{m.nl, e.m4, e.s4, e.s2};
    int n = __popc(e.m4);
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][0] = n;
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int k = 0; k < TK_BLOCK / 64; ++k) tot += sh[k][0];
        tile_nl[blockIdx.x] = s.v_total; tile_bytes[blockIdx.x] = tot;
    }
}

// launch 2 of 3, one workgroup: tile_nl / tile_bytes -> newlines / column-5 bytes in front of every tile (in place); totals and status -> meta.
// 8,192 tiles (64 MB of text) per round: their counts come in with coalesced loads (all in flight at once) and wait in LDS, a thread sums
// its eight consecutive ones, the workgroup scans the 1,024 sums with wave shuffles, the prefixes go out as they are formed.
constexpr int TK_SCAN_ROUND = 8192;
__global__ __launch_bounds__(1024) void k_tok_scan(int64_t* __restrict__ tile_nl, int64_t* __restrict__ tile_bytes, int64_t n_tiles,
                                                    const int64_t* __restrict__ ws_meta, int64_t cap_cols, int64_t cap_bytes,
                                                    int64_t* __restrict__ col_off, int64_t* __restrict__ meta)
{
    __shared__ int vn[TK_SCAN_ROUND], vb[TK_SCAN_ROUND];          // (a tile holds at most 8,192 of either)
    __shared__ long long sh_v[16];
    __shared__ int sh_s[16];
    const int tid = threadIdx.x;
    long long carry_n = 0, carry_b = 0;
    for (int64_t base = 0; base < n_tiles; base += TK_SCAN_ROUND) {
        const int cnt = (int)(n_tiles - base < TK_SCAN_ROUND ? n_tiles - base : TK_SCAN_ROUND);
#pragma unroll
        for (int k = 0; k < TK_SCAN_ROUND / 1024; ++k) {
            const int i = k * 1024 + tid;
            vn[i] = i < cnt ? (int)tile_nl[base + i] : 0;
            vb[i] = i < cnt ? (int)tile_bytes[base + i] : 0;
        }
        __syncthreads();
        constexpr int PER = TK_SCAN_ROUND / 1024;
        long long sn = 0, sb = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { sn += vn[PER * tid + k]; sb += vb[PER * tid + k]; }
        const WgScan wn = tk_wg_scan1024(sn, 0, sh_v, sh_s);
        const WgScan wb = tk_wg_scan1024(sb, 0, sh_v, sh_s);
        long long rn = carry_n + wn.v_excl, rb = carry_b + wb.v_excl;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = PER * tid + k;
            if (i < cnt) { tile_nl[base + i] = rn; tile_bytes[base + i] = rb; }
            rn += vn[i]; rb += vb[i];
        }
        carry_n += wn.v_total; carry_b += wb.v_total;
        __syncthreads();
    }
    if (tid == 0) {
        int64_t status = ws_meta[2];
        if (carry_n > cap_cols || carry_b > cap_bytes) status |= TOK_ERANGE;
        else col_off[carry_n] = carry_b;
        meta[0] = carry_n; meta[1] = carry_b; meta[2] = status; meta[3] = 0;
    }
}

// launch 3 of 3: per tile again, now with the line index and the output offset of its first byte: positions and reference bytes of the
// lines whose token 1 starts here, column offsets of the tokens 4 that start here, their bytes compacted through LDS into `bases`
__global__ __launch_bounds__(TK_BLOCK) void k_tok_emit(TokText t, const int64_t* __restrict__ tile_nl, const int64_t* __restrict__ tile_bytes,
                                                        const uint4* __restrict__ chunk_masks, const uint8_t* __restrict__ chr_seq, int64_t chr_len,
                                                        int64_t cap_cols, int64_t cap_bytes, int64_t* __restrict__ pos, uint8_t* __restrict__ ref,
                                                        int64_t* __restrict__ col_off, uint8_t* __restrict__ bases, int64_t* __restrict__ ws_meta)
{
    __shared__ int sh[TK_BLOCK / 64][2];
    __shared__ __attribute__((aligned(16))) uint8_t txt[TK_TILE];
    __shared__ __attribute__((aligned(16))) uint8_t cbuf[TK_TILE + 32];
    const int tid = threadIdx.x;
    const int64_t tile0 = (int64_t)blockIdx.x * TK_TILE, p0 = tile0 + tid * TK_CHUNK;
    uint32_t w[8];
    tk_load_words(t, p0, w);
    tk_stage_text(txt, w);
    const uint4 cm = chunk_masks[(int64_t)blockIdx.x * TK_BLOCK + tid];
    const uint32_t nlm = cm.x;
    const TokEvents e{cm.y, cm.z, cm.w, 0u};
    // one scan, two sums: newlines (high half) and column-5 bytes (low half) in front of this chunk inside the tile
    const BlockScan s2 = tk_block_scan((__popc(nlm) << 16) | __popc(e.m4), 0, sh);            // (its barriers publish txt)
    const int64_t line0 = tile_nl[blockIdx.x] + (s2.v_excl >> 16);
    const int r0 = s2.v_excl & 0xffff, tile_cnt = s2.v_total & 0xffff;
    const int64_t out0 = tile_bytes[blockIdx.x];
    const int mis = (int)(out0 & 15);
    uint32_t err = 0;
    for (uint32_t s2 = e.s2; s2; s2 &= s2 - 1) {
        const int b = __ffs(s2) - 1;
        const int64_t line = line0 + __popc(nlm & ((1u << b) - 1u));
        const int64_t v = tk_atoll(t, txt, tile0, p0 + b);
        if (line < cap_cols) {
            pos[line] = v;
            if (ref) {
                if (v >= 1 && v <= chr_len) ref[line] = chr_seq[v - 1];
                else { ref[line] = 'N'; err |= TOK_EPOS; }
            }
        }
    }
    if (err) atomicOr(reinterpret_cast<unsigned long long*>(ws_meta + 3), (unsigned long long)err);
    for (uint32_t s4 = e.s4; s4; s4 &= s4 - 1) {
        const int b = __ffs(s4) - 1;
        const uint32_t below = (1u << b) - 1u;
        const int64_t line = line0 + __popc(nlm & below);
        if (line < cap_cols) col_off[line] = out0 + r0 + __popc(e.m4 & below);
    }
    {
        // (the bytes come from the tile's copy in LDS: picking byte b out of the eight registers is a chain of selects per byte)
        int r = mis + r0;
        const uint8_t* mine = txt + tid * TK_CHUNK;
        for (uint32_t rest = e.m4; rest; rest &= rest - 1) cbuf[r++] = mine[__ffs(rest) - 1];
    }
    __syncthreads();
    // 16-byte pieces of bases[out0 - mis, out0 + tile_cnt): whole pieces as one store, the two ragged ones byte by byte
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
}

// a position outside the reference is found by launch 3, behind the scan that wrote meta: one thread folds it in
// (it also re-arms the two status words for the next call: no memset in front of every call)
__global__ void k_tok_status(int64_t* __restrict__ ws_meta, int64_t* __restrict__ meta)
{
    if (ws_meta[3]) meta[2] |= ws_meta[3];
    ws_meta[2] = 0; ws_meta[3] = 0;
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
    const size_t need = (size_t)n_tiles * (8 + 8 + 16 * TK_BLOCK) + 64 + 256;      // per tile: newlines, bytes, 16 bytes of masks per thread (the chained scan: 16 per tile)
    if (ctx->tok_ws_bytes >= need) return NSNP_OK;
    NSNP_HIP(ctx, hipStreamSynchronize(s));
    if (ctx->tok_ws) (void)hipFree(ctx->tok_ws);
    ctx->tok_ws = nullptr; ctx->tok_ws_bytes = 0;
    const size_t want = need + need / 4;
    NSNP_HIP(ctx, hipMalloc(&ctx->tok_ws, want));
    ctx->tok_ws_bytes = want;
    NSNP_HIP(ctx, hipMemsetAsync(ctx->tok_ws, 0, 64, s));            // the status words start at zero; every call leaves them so (k_tok_status)
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
    if (text_len == 0) {
        NSNP_HIP(ctx, hipMemsetAsync(meta, 0, 4 * sizeof(int64_t), s));
        NSNP_HIP(ctx, hipMemsetAsync(col_off, 0, sizeof(int64_t), s));
        return NSNP_OK;
    }
    const int mis = (int)((uintptr_t)text & 15);
    TokText t;
    t.base = text - mis; t.lo = mis; t.hi = mis + text_len;
    const int64_t n_tiles = NSNP_CDIV(t.hi + 1, (int64_t)TK_TILE);
    int rc = tok_reserve(ctx, n_tiles, s);
    if (rc != NSNP_OK) return rc;
    uint8_t* ws = (uint8_t*)ctx->tok_ws;
    if (ctx->tok_fused) {
        // one launch: its descriptors (16 bytes per tile) zeroed in front of it
        TokDesc* desc = (TokDesc*)(ws + 64);
        NSNP_HIP(ctx, hipMemsetAsync(desc, 0, (size_t)n_tiles * sizeof(TokDesc), s));
        hipLaunchKernelGGL(k_tok_fused, dim3((unsigned)n_tiles), dim3(TK_BLOCK), 0, s, t, (long long)n_tiles, desc, chr_seq, chr_len, cap_cols, cap_bytes,
                           pos, ref, col_off, bases, meta);
        NSNP_HIP(ctx, hipGetLastError());
        return NSNP_OK;
    }
    int64_t* ws_meta = (int64_t*)ws;                               // [2] status bits of launch 1, [3] of launch 3
    int64_t* tile_nl = (int64_t*)(ws + 64);
    int64_t* tile_bytes = tile_nl + n_tiles;
    uint4* chunk_masks = (uint4*)(tile_bytes + n_tiles);           // (64 + 16 n_tiles bytes in: 16-byte aligned)
    hipLaunchKernelGGL(k_tok_count, dim3((unsigned)n_tiles), dim3(TK_BLOCK), 0, s, t, tile_nl, tile_bytes, chunk_masks, ws_meta);
    hipLaunchKernelGGL(k_tok_scan, dim3(1), dim3(1024), 0, s, tile_nl, tile_bytes, n_tiles, (const int64_t*)ws_meta, cap_cols, cap_bytes, col_off, meta);
    hipLaunchKernelGGL(k_tok_emit, dim3((unsigned)n_tiles), dim3(TK_BLOCK), 0, s, t, (const int64_t*)tile_nl, (const int64_t*)tile_bytes,
                       (const uint4*)chunk_masks, chr_seq, chr_len, cap_cols, cap_bytes, pos, ref, col_off, bases, ws_meta);
    hipLaunchKernelGGL(k_tok_status, dim3(1), dim3(1), 0, s, ws_meta, meta);
    NSNP_HIP(ctx, hipGetLastError());
    return NSNP_OK;
}
