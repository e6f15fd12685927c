// K4: BM25 over CSR postings. Replaces BM25Okapi(tokenized_corpus).get_scores(tokenized_query) and the /max
// normalisation of /root/reference/rag/retrieval.py:324-347 (rank-bm25 0.2.2 semantics, float64).
//
// Integer/gather work, not a GEMM (89 % of the posting reads hit L2: frequent terms are shared by the batch's queries):
//   load   : per-posting impact w = idf * tf*(k1+1) / (tf + k1*(1 - b + b*dl/avgdl)) precomputed once in float64 with
//            rank-bm25's operation order -> postings are (doc int32, w float64): 12 B each, no doc_len gather, no multiply
//            later. A per-term BRACKET table (first posting at every 2^shift-th document, built once on the GPU) replaces
//            the two 20-step dependent binary searches per (block, token) that dominated the first version (36 ms/batch).
//            The table is two-level in the vocabulary: a term's bracket width is the smallest power of two >= the
//            2048-document range for which the table stays under df/4 entries, so long posting lists get the direct
//            per-range row (zero search steps, as before), the middle of the distribution coarser rows (a 1-3 step
//            search inside one bracket), and terms with fewer than 8 postings no row at all (a <= 3 step search of the
//            whole list). Table bytes <= nnz bytes = 1/12 of the postings for ANY vocabulary (r2's dense
//            V x n_ranges table was 122 GB for 5M terms on a 12.5M-document shard).
//   score  : one 256-thread workgroup per (query, 2048-doc range). The range's float64 accumulators live in LDS (20.5 KiB
//            with the scratch: SEVEN workgroups per CU, so one's posting round trips and token barriers hide behind the
//            others' adds; hybrid batch of 1024 on one box: 16384 docs x 1024 threads = one workgroup per CU 132.8 k q/s,
//            8192 x 1024 138.2 k, 4096 x 512 144.8 k, 2048 x 256 151.2 k); for each query token IN ORDER the block adds the
//            term's impacts of this doc range (docs are unique inside one posting list -> one read-add-write per
//            accumulator per token, and per-document summation order is the query-token order, exactly as `score += ...`
//            in get_scores -> bit-identical float64). 4 consecutive postings per thread per trip, the next trip in flight
//            across the token barrier.
//            Algorithmic traffic per query = sum over tokens of df*12 B; accumulators never touch HBM.
//   select : STAGED. The first BM_FIRST_RANGES ranges get an exact per-range top-k (8-pass radix select on order-
//            preserving keys, ties -> lower doc id, i.e. Python's stable sort); bm25_tau_kernel takes the k-th best key
//            over them (a lower bound of the global k-th); every other range only compacts its keys >= tau (exact select
//            as the fallback when more than k survive). Then a per-query merge of the partial lists.
#include "common.h"

typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));          // posting segments start at any posting
typedef double double2u __attribute__((ext_vector_type(2), aligned(8)));
#ifndef BM_RANGE            // -DBM_RANGE / -DBM_THREADS: variant builds by tools/bm25_variant_build.sh
#define BM_RANGE 2048
#endif
#ifndef BM_THREADS
#define BM_THREADS 256
#endif
#define BM_SEG (BM_RANGE / BM_THREADS)      // 8 contiguous docs per thread
// accumulator i lives at LDS double i + i/32 (one pad double per 32 docs). The padding dates from 32 docs per thread, where it
// made the per-thread segment reads of the select conflict-free; with 8 docs per thread it still spreads the lanes (lane t
// starts at bank pair 8(t%4) + t/4: at most 2-way instead of 8-way). The adds of the token loop fall on random banks either way.
#define SC_IDX(i) ((i) + ((i) >> 5))
// the same place as an LDS byte offset, from an UNSIGNED index: shift, add-and-shift (the pointer form &sc[SC_IDX(i)] costs the
// token loop two more vector instructions per posting, and that loop is bound by vector-instruction issue - DESIGN 4.2)
#define SC_OFF(u) ((((uint32_t)(u)) + (((uint32_t)(u)) >> 5)) << 3)
#define BM_SC_DOUBLES (BM_RANGE + BM_RANGE / 32)
#define BM_LDS_BYTES (BM_SC_DOUBLES * 8 + 4096)

// everything a (workgroup, token) needs to know about the token's term: ONE 32-byte load
struct __attribute__((aligned(32))) bm_term_meta {
    int64_t post;       // first posting of the term (indptr[t])
    int64_t tab;        // first entry of the term's bracket table in range_tab
    double idf;
    int32_t df;         // number of postings
    int32_t shift;      // log2(documents per bracket); BM_NO_TAB: no table, the bracket is the whole list
};
#define BM_NO_TAB 63
#define BM_TAB_DIV 4    // a term's table may take df / 4 entries (4 B each): table bytes <= nnz bytes = 1/12 of the postings
#define BM_RANGE_LOG2 (31 - __builtin_clz(BM_RANGE))
static_assert((BM_RANGE & (BM_RANGE - 1)) == 0, "BM_RANGE must be a power of two");

struct bm_plan_meta;
struct rag_bm25_index {
    int64_t n_docs = 0, n_terms = 0, nnz = 0;
    int64_t* indptr = nullptr;
    int32_t* doc = nullptr;
    double* w = nullptr;
    double* idf = nullptr;
    // OPTIONAL compact form (option bm25_packed at load time): 4-byte postings (doc & (BM_RANGE - 1)) | code << BM_RANGE_LOG2, code =
    // tf rank * n_dl + length rank, and the table g[code] = tf (k1+1) / (tf + k1 (1 - b + b dl / avgdl)) shared by all terms
    // (impact = idf * g, the same float64 product): 8 B per posting resident instead of 12 (`doc` stays for the plan kernel's
    // searches, `w` is not built) and a third of the streamed bytes - but the scoring loop gets SLOWER (1M documents, 1024 queries:
    // 4.22 against 3.36 ms): four 8-byte table gathers per chunk cost the texture path more than the two coalesced impact loads
    // they replace. The default therefore streams (doc i32, impact f64).
    uint32_t* packed = nullptr;
    double* gtab = nullptr;
    int64_t n_codes = 0;
    bm_term_meta* meta = nullptr;      // [n_terms]
    int32_t* range_tab = nullptr;      // concatenated per-term tables: entry c = first posting (rel. to meta.post) with doc >= c << shift
    int64_t tab_entries = 0;
    int n_ranges = 0;
    uint64_t* ws_key = nullptr;        // per-range partial top-k workspace for the device entry point
    uint32_t* ws_row = nullptr;
    size_t ws_entries = 0;
    uint64_t* ws_tau = nullptr;        // [ws_tau_q] first-stage threshold per query
    int ws_tau_q = 0;
    int* ws_cnt = nullptr;             // [ws_cnt_n] valid entries per (query, range) partial list
    size_t ws_cnt_n = 0;
    uint64_t* ws_run_key = nullptr;    // [ws_run_n] running top-k of every query across the threshold stages
    uint32_t* ws_run_row = nullptr;
    size_t ws_run_n = 0;
    int32_t* ws_plan_off = nullptr;    // per-call plan of the device entry points (bm25_plan_kernel), grown on demand
    bm_plan_meta* ws_plan_meta = nullptr;
    size_t ws_plan_entries = 0;
    int ws_plan_q = 0;
    int plan_t = 64;                   // planned token slots per query of the CURRENT call (bm25_pick_plan_t), <= BM_PLAN_T
    double avgdl = 0, k1 = 1.5, b = 0.75;
    int normalize = 1;                 // 0: top-k scores stay raw (row-sharded search divides by the GLOBAL max after the merge)
};

__global__ void bm25_weights_kernel(const int64_t* __restrict__ indptr, const int32_t* __restrict__ doc,
                                    const int32_t* __restrict__ tf, const int32_t* __restrict__ doc_len, int64_t nnz,
                                    double avgdl, double k1, double b, const double* __restrict__ idf, int64_t n_terms,
                                    double* __restrict__ w) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    // term of posting p: last t with indptr[t] <= p
    int64_t lo = 0, hi = n_terms;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= p) lo = mid; else hi = mid;
    }
    const double f = (double)tf[p];
    const double dl = (double)doc_len[doc[p]];
    // q_freq * (k1 + 1) / (q_freq + k1 * (1 - b + b * doc_len / avgdl))   -- same association as rank-bm25
    const double num = f * (k1 + 1.0);
    const double t1 = (b * dl) / avgdl;
    const double t2 = (1.0 - b) + t1;
    const double den = f + k1 * t2;
    // the impact is stored already multiplied by the term's idf: `idf * (...)` is the product rank-bm25 adds to the score, so
    // the scoring loop is a pure load + add (one float64 multiply and one LDS lookup fewer per posting)
    w[p] = idf[lo] * (num / den);
}

// packed postings (see rag_bm25_index): one thread per posting
__global__ void bm25_pack_kernel(const int32_t* __restrict__ doc, const int32_t* __restrict__ tf, const uint32_t* __restrict__ tf_rank,
                                 const uint32_t* __restrict__ dl_rank_of_doc, int64_t nnz, uint32_t n_dl, uint32_t* __restrict__ packed) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    const int32_t d = doc[p];
    packed[p] = ((uint32_t)d & (BM_RANGE - 1)) | ((tf_rank[tf[p]] * n_dl + dl_rank_of_doc[d]) << BM_RANGE_LOG2);
}

// g[ti * n_dl + di] for the ti-th distinct term frequency and the di-th distinct document length - the second factor of
// rank-bm25's `idf * (q_freq * (k1 + 1) / (q_freq + k1 * (1 - b + b * doc_len / avgdl)))`, same association as bm25_weights_kernel
__global__ void bm25_gtab_kernel(const double* __restrict__ tf_values, const double* __restrict__ dl_values, int64_t n_codes, uint32_t n_dl,
                                 double avgdl, double k1, double b, double* __restrict__ gtab) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_codes) return;
    const double f = tf_values[c / n_dl], dl = dl_values[c % n_dl];
    const double num = f * (k1 + 1.0);
    const double t1 = (b * dl) / avgdl;
    const double t2 = (1.0 - b) + t1;
    const double den = f + k1 * t2;
    gtab[c] = num / den;
}

__device__ __forceinline__ int64_t lower_bound_doc(const int32_t* __restrict__ doc, int64_t lo, int64_t hi, int target) {
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (doc[mid] < target) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Host-side plan of the bracket tables: shift and entry count per term (see the header comment). n_pad = documents rounded
// up to a whole range; a table with shift g has ceil(n_pad / 2^g) + 1 entries (the last one = df).
static inline int64_t bm_tab_entries(int64_t n_pad, int g) { return ((n_pad + ((int64_t)1 << g) - 1) >> g) + 1; }
static inline int bm_plan_term(int64_t df, int64_t n_pad, int64_t* entries_out) {
    const int64_t budget = df / BM_TAB_DIV;
    int g = BM_RANGE_LOG2;
    while (bm_tab_entries(n_pad, g) > budget && bm_tab_entries(n_pad, g) > 2) ++g;
    const int64_t e = bm_tab_entries(n_pad, g);
    if (e > budget || e <= 2) { *entries_out = 0; return BM_NO_TAB; }      // one bracket = the whole list: no table needed
    *entries_out = e;
    return g;
}

// one thread per table entry: term = last t whose table starts at or before entry i (terms without a table share their
// start with the next term, so the LAST such t is the owner)
__global__ void bm25_range_table_kernel(const bm_term_meta* __restrict__ meta, const int32_t* __restrict__ doc, int64_t n_terms,
                                        int64_t n_entries, int32_t* __restrict__ range_tab) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_entries) return;
    int64_t lo = 0, hi = n_terms;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (meta[mid].tab <= i) lo = mid; else hi = mid;
    }
    const bm_term_meta m = meta[lo];
    const int64_t target = (i - m.tab) << m.shift;
    range_tab[i] = target > 0x7fffffff ? m.df : (int32_t)(lower_bound_doc(doc, m.post, m.post + m.df, (int)target) - m.post);
}

// lower bound of `tg` in the ascending doc list dl[lo, hi): binary steps while the bracket is longer than 8 postings, then ONE
// round trip: the (up to) 8 remaining doc ids come as two 16-byte loads and are counted (the arrays carry 8 postings of padding)
__device__ __forceinline__ int bm_search(const int32_t* __restrict__ dl, int lo, int hi, int64_t tg) {
    while (hi - lo > 8) {
        const int mid = (lo + hi) >> 1;
        if (dl[mid] < tg) lo = mid + 1; else hi = mid;
    }
    if (hi > lo) {
        const int4u v0 = *reinterpret_cast<const int4u*>(dl + lo), v1 = *reinterpret_cast<const int4u*>(dl + lo + 4);
        const int n = hi - lo;
        int c = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            c += (j < n) && (v0[j] < tg);
            c += (j + 4 < n) && (v1[j] < tg);
        }
        lo += c;
    }
    return lo;
}

// bracket [lo, hi] of the term's table that holds lower_bound(tg); a target ON a bracket edge is the entry itself (always the
// case for the direct per-range rows of long lists); no table: the whole list
__device__ __forceinline__ void bm_bracket(const bm_term_meta& m, const int32_t* __restrict__ range_tab, int64_t tg, int& lo, int& hi) {
    lo = 0;
    hi = m.df;
    if (m.shift != BM_NO_TAB) {
        const int32_t* tb = range_tab + m.tab;
        const int64_t c = tg >> m.shift;
        lo = tb[c];
        hi = (tg & (((int64_t)1 << m.shift) - 1)) ? tb[c + 1] : lo;
    }
}

// ---- per-call PLAN: the posting offsets of every (query token, doc range) ------------------------------------------------
// plan_off[q][slot][r] = first posting (relative to the term's list) with doc >= r * BM_RANGE, r = 0 .. n_ranges, for the first
// BM_PLAN_T tokens of query q; plan_meta[q][slot] = the term's first posting and idf (idf 0: unknown token). One thread per
// (query, range edge), looping over the query's tokens: bracket lookup + in-bracket search, all independent - the searches the
// two-level table needs run here, massively parallel, instead of at the head of every scoring workgroup's dependent chain.
// The scoring kernel then reads two adjacent plan entries per token, exactly as it read r2's dense per-term table.
// The plan holds plan_t <= BM_PLAN_T token slots per query, chosen per call so that the table stays under BM_PLAN_BUDGET bytes
// (64 slots x 6,104 ranges x 4 B = 1.56 MB per query on a 12.5M-document shard whatever the queries' real token counts,
// 1.6 GB at 1,024 queries); tokens past the plan are searched inside the scoring kernel, so results do not depend on plan_t.
#define BM_PLAN_T 64
#define BM_PLAN_BUDGET ((size_t)512 << 20)
struct __attribute__((aligned(16))) bm_plan_meta { int64_t post; double idf; };

__global__ __launch_bounds__(256) void bm25_plan_kernel(const bm_term_meta* __restrict__ meta, const int32_t* __restrict__ doc,
                                                         const int32_t* __restrict__ range_tab, int n_ranges, int64_t n_terms,
                                                         const int32_t* __restrict__ term_ptr, const int32_t* __restrict__ terms,
                                                         int32_t* __restrict__ plan_off, bm_plan_meta* __restrict__ plan_meta, int plan_t) {
    const int q = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
    const int t0 = term_ptr[q], nt = min(plan_t, term_ptr[q + 1] - t0);
    const int64_t tg = (int64_t)r * BM_RANGE;
    for (int s = 0; s < nt; ++s) {
        const int t = terms[t0 + s];
        const bool ok = t >= 0 && t < n_terms;               // out-of-vocabulary token: idf.get(q) is None -> 0
        bm_term_meta m = {0, 0, 0.0, 0, BM_NO_TAB};
        if (ok) m = meta[t];
        if (r == 0) plan_meta[(size_t)q * plan_t + s] = {m.post, ok ? m.idf : 0.0};
        if (r <= n_ranges) {
            int lo, hi;
            bm_bracket(m, range_tab, tg, lo, hi);
            plan_off[((size_t)q * plan_t + s) * (n_ranges + 1) + r] = bm_search(doc + m.post, lo, hi, tg);
        }
    }
}

// Launch geometry of the scoring kernel: a 1-D grid over (range of this launch, query). The hardware hands consecutive workgroup
// ids to the 8 XCDs in turn, each XCD has its own 4 MiB L2, and one range's postings are ~2 MB: with the queries of a column
// running side by side on one XCD, the posting segments the queries share (frequent terms: most of the bytes) are served by
// that L2. n_groups splits the queries of a range into several columns when a launch has few ranges (the opening stages), so
// that every XCD owns >= 16 columns or so and the XCDs finish together. Few queries (qgroup_len = 0): range-major, the ranges
// of one query spread over the chip. The id -> XCD rule is a performance assumption only; any placement gives the same result.
struct bm_grid { unsigned blocks; int nr_l, n_queries, n_groups, qgroup_len, plan_t; };
// mode 1 (all-document scores) can also hand the index-level linear fusion what it needs, in the same pass over the accumulators:
// a float32 copy of the raw scores [q][ld] (the emission operand of the fused dense search) and the per-query maximum over the
// tenant's documents as an orderable key (atomicMax: order-independent, so the result is the bits a sequential max gives).
struct bm_dense_extra { float* raw32; int64_t ld; unsigned long long* max_key; };
static bm_grid bm_make_grid(int nr_l, int Q, int linear) {
    bm_grid g{(unsigned)((int64_t)nr_l * Q), nr_l, Q, 1, 0, BM_PLAN_T};
    if (linear || Q < 128) return g;
    int G = 1;
    while (nr_l * G < 128 && G < 8 && Q / (2 * G) >= 64) G <<= 1;
    const int ql = (Q + G - 1) / G;
    const int64_t per_xcd = ((int64_t)nr_l * G + 7) / 8;
    g.blocks = (unsigned)(8 * per_xcd * ql);
    g.n_groups = G;
    g.qgroup_len = ql;
    return g;
}

// One workgroup per (query, doc range).
// mode 0: per-range top-k partials; mode 1: dense scores out[q][doc]
// PACKED (option bm25_packed, off by default - see rag_bm25_index): the scoring loop streams 4-byte postings (document number
// inside its 2048-document range | code of its (tf, doc length) pair) and looks the impact factor g(tf, dl) up in a table shared
// by all terms. Default: (doc i32, impact f64) = 12 bytes per posting.
// (SGPRs capped at 96: 256-thread workgroups are admitted 7 per CU up to 96 scalar registers, 6 from 97 on - MI355X_MICROARCH.md)
template <bool PACKED>
__global__ __launch_bounds__(BM_THREADS) __attribute__((amdgpu_num_sgpr(96))) void bm25_range_kernel(const bm_term_meta* __restrict__ meta, const int32_t* __restrict__ doc,
                                                                 const double* __restrict__ w, const uint32_t* __restrict__ packed,
                                                                 const double* __restrict__ gtab,
                                                                 const int32_t* __restrict__ range_tab, int n_ranges,
                                                                 const int32_t* __restrict__ term_ptr, const int32_t* __restrict__ terms,
                                                                 int64_t n_docs, int64_t n_terms, int k, int mode,
                                                                 double* __restrict__ dense_out, uint64_t* __restrict__ part_key,
                                                                 uint32_t* __restrict__ part_row, int range_begin,
                                                                 const uint64_t* __restrict__ tau_key, int* __restrict__ part_cnt,
                                                                 const int32_t* __restrict__ tenants, int tenant,
                                                                 const int32_t* __restrict__ plan_off, const bm_plan_meta* __restrict__ plan_meta,
                                                                 const bm_grid gm, const bm_dense_extra dx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // workgroup -> (range, query), see bm_make_grid: XCD x (= workgroup id % 8) walks its columns (range, query group) one after the
    // other, every query of the group on the SAME range side by side, so a range's postings are fetched into that XCD's L2 once
    // per column instead of once per query that holds the term.
    int r, q;
    if (gm.qgroup_len == 0) {
        r = range_begin + (int)(blockIdx.x % (unsigned)gm.nr_l);
        q = (int)(blockIdx.x / (unsigned)gm.nr_l);
    } else {
        const unsigned x = blockIdx.x & 7u, s_ = blockIdx.x >> 3;
        const unsigned col = x + 8u * (s_ / (unsigned)gm.qgroup_len), qi = s_ % (unsigned)gm.qgroup_len;
        const int rl = (int)(col / (unsigned)gm.n_groups);
        q = (int)(col % (unsigned)gm.n_groups) * gm.qgroup_len + (int)qi;
        if (rl >= gm.nr_l || q >= gm.n_queries) return;
        r = range_begin + rl;
    }
    // the query's threshold (thresholded stages) is fetched up front: at the compaction step it would be an exposed global
    // round trip for every workgroup
    const uint64_t tk = tau_key != nullptr ? tau_key[q] : 0ull;
    double* sc = reinterpret_cast<double*>(smem);                       // [BM_RANGE]
    int* hist = reinterpret_cast<int*>(smem + BM_SC_DOUBLES * 8);       // [256]
    int* wsum = hist + 256;                                             // [16] scratch
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int64_t base = (int64_t)r * BM_RANGE;
    const int lim = (int)min((int64_t)BM_RANGE, n_docs - base);
    // per-token metadata (idf, posting sub-range of this doc range) is fetched for up to 64 tokens IN PARALLEL into
    // LDS first: fetched inside the token loop it was ~4 dependent global round trips per token per block. The first batch's
    // loads (the planned tokens: two adjacent plan entries each) are issued BEFORE the accumulators are cleared, so that the
    // clearing runs under their round trip, and one barrier covers both.
    double* m_idf = reinterpret_cast<double*>(wsum + 16);              // [64]
    int64_t* m_a = reinterpret_cast<int64_t*>(m_idf + 64);             // [64]
    int* m_n = reinterpret_cast<int*>(m_a + 64);                       // [64]
    const int t0 = term_ptr[q], t1 = term_ptr[q + 1];
    double f_first = 0.0;
    int64_t a_first = 0;
    int n_first = 0;
    if (tid < min(gm.plan_t, t1 - t0)) {
        const bm_plan_meta pm = plan_meta[(size_t)q * gm.plan_t + tid];
        const int32_t* po = plan_off + ((size_t)q * gm.plan_t + tid) * (n_ranges + 1) + r;
        const int o0 = po[0], o1 = po[1];
        f_first = pm.idf;
        a_first = pm.post + o0;
        n_first = o1 - o0;
    }
    for (int i = tid; i < BM_SC_DOUBLES; i += BM_THREADS) sc[i] = 0.0;
    if (t1 <= t0) __syncthreads();                           // no token at all: the select below still needs the cleared array
    for (int tb = t0; tb < t1; tb += 64) {
        const int nb = min(64, t1 - tb);
        if (tid < nb) {
            double f = 0.0;
            int64_t a = 0;
            int n = 0;
            if (tb == t0 && tid < gm.plan_t) {               // planned tokens (at most one batch)
                f = f_first;
                a = a_first;
                n = n_first;
            } else {                                         // tokens past the plan (more than plan_t tokens in the query): search here
                const int t = terms[tb + tid];
                if (t >= 0 && t < n_terms) {
                    const bm_term_meta m = meta[t];
                    int lo0, hi0, lo1, hi1;
                    bm_bracket(m, range_tab, base, lo0, hi0);
                    bm_bracket(m, range_tab, base + BM_RANGE, lo1, hi1);
                    const int s0 = bm_search(doc + m.post, lo0, hi0, base);
                    const int e0 = bm_search(doc + m.post, max(lo1, s0), hi1, base + BM_RANGE);
                    f = m.idf;
                    a = m.post + s0;
                    n = e0 - s0;
                }
            }
            m_idf[tid] = f;
            m_a[tid] = a;
            m_n[tid] = f == 0.0 ? 0 : n;                     // (idf or 0) * x == 0: adds nothing
        }
        __syncthreads();
        // Docs are unique inside one posting list: exactly one add per accumulator per token, so a plain LDS read-add-write
        // is race-free inside a token (a no-return ds_add_f64 gives the same bits and was 2 % slower on the hybrid batch:
        // the float64 LDS atomic runs well below the rate of a b64 read + write); tokens are separated by a barrier so that
        // the per-document summation order is the token order (bit-identical to numpy). A (token, range) segment is only a few
        // postings per thread, so the loop is bound by one global round trip per token unless the NEXT chunk (4
        // consecutive postings per thread = 4 x BM_THREADS per block, possibly of the next token) is already in flight while the
        // current one is added: two register sets ping-pong, loads are unconditional (the arrays carry 4 postings of
        // padding) so the compiler keeps counted vmcnt waits, and the token barrier is a raw s_barrier behind
        // lgkmcnt(0) — __syncthreads would drain vmcnt and the prefetch with it.
        // The walk over (token, chunk) is the same in every thread. Each wave keeps the batch's token metadata in one VGPR
        // set (lane t = token t) and reads it with v_readlane at a scalar index: the three dependent LDS round trips per trip
        // (segment length for the step, base + length for the address, length for the tail test) become scalar moves.
        const int my_n = m_n[lane];
        const int64_t my_a = m_a[lane];
        const int my_a_lo = (int)(uint32_t)my_a, my_a_hi = (int)(my_a >> 32);
#define TOK_N(TI) __builtin_amdgcn_readlane(my_n, TI)
#define TOK_A(TI) (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane(my_a_hi, TI) << 32) | (uint32_t)__builtin_amdgcn_readlane(my_a_lo, TI))
        int ti = 0, c = 0;
        while (ti < nb && TOK_N(ti) == 0) ++ti;
#define BM_NEXT(TI, C, NTI, NC)                                                                    \
            NTI = TI; NC = C + 1;                                                                  \
            if (NC * 4 * BM_THREADS >= TOK_N(TI)) {                                                \
                NC = 0;                                                                            \
                ++NTI;                                                                             \
                while (NTI < nb && TOK_N(NTI) == 0) ++NTI;                                         \
            }
#define BM_TOKEN_BARRIER asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
        if constexpr (PACKED) {
            // 4-byte postings: u = (doc - base) | code << 11; the impact is idf(token) * gtab[code], one float64 multiply - the same
            // product the 12-byte form stored (`idf * (num / den)`, rank-bm25's association), so the sums are bit-identical.
            // Three chunks are in flight per thread: the postings of chunk i + 2, the table values of chunk i + 1 (they need that
            // chunk's postings), the adds of chunk i.
            const double my_f = m_idf[lane];
            const int my_f_lo = (int)(uint32_t)__builtin_bit_cast(uint64_t, my_f), my_f_hi = (int)(__builtin_bit_cast(uint64_t, my_f) >> 32);
#define TOK_IDF(TI) __builtin_bit_cast(double, ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(my_f_hi, TI) << 32) | (uint32_t)__builtin_amdgcn_readlane(my_f_lo, TI))
#define BMP_LOAD(TI, C, U)                                                                         \
            {                                                                                      \
                const int64_t o_ = TOK_A(TI) + min((C) * 4 * BM_THREADS + tid * 4, max(TOK_N(TI) - 1, 0)); \
                U = *reinterpret_cast<const int4u*>(packed + o_);                                  \
            }
#define BMP_GATHER(U, G)                                                                           \
            {                                                                                      \
                G[0] = gtab[(uint32_t)U[0] >> BM_RANGE_LOG2]; G[1] = gtab[(uint32_t)U[1] >> BM_RANGE_LOG2]; \
                G[2] = gtab[(uint32_t)U[2] >> BM_RANGE_LOG2]; G[3] = gtab[(uint32_t)U[3] >> BM_RANGE_LOG2]; \
            }
#define BMP_ADD(TI, C, U, G)                                                                       \
            {                                                                                      \
                const int p_ = (C) * 4 * BM_THREADS + tid * 4, n_ = TOK_N(TI);                     \
                if (p_ < n_) {                                                                     \
                    const double f_ = TOK_IDF(TI);                                                 \
                    double* a0_ = &sc[SC_IDX(U[0] & (BM_RANGE - 1))];                              \
                    double* a1_ = p_ + 1 < n_ ? &sc[SC_IDX(U[1] & (BM_RANGE - 1))] : &sc[BM_SC_DOUBLES - 1]; \
                    double* a2_ = p_ + 2 < n_ ? &sc[SC_IDX(U[2] & (BM_RANGE - 1))] : &sc[BM_SC_DOUBLES - 2]; \
                    double* a3_ = p_ + 3 < n_ ? &sc[SC_IDX(U[3] & (BM_RANGE - 1))] : &sc[BM_SC_DOUBLES - 3]; \
                    const double v0_ = *a0_, v1_ = *a1_, v2_ = *a2_, v3_ = *a3_;                   \
                    *a0_ = v0_ + f_ * G[0];                                                        \
                    if (p_ + 1 < n_) *a1_ = v1_ + f_ * G[1];                                       \
                    if (p_ + 2 < n_) *a2_ = v2_ + f_ * G[2];                                       \
                    if (p_ + 3 < n_) *a3_ = v3_ + f_ * G[3];                                       \
                }                                                                                  \
            }
            // one turn of the pipeline: chunk X is added, chunk Y (postings in flight) gets its table values requested, the
            // postings of the chunk after Y are requested into Z. (tx, cx) / (ty, cy) walk the (token, chunk) sequence.
#define BMP_TURN(UX, GX, UY, GY, UZ)                                                               \
            {                                                                                      \
                int tz_ = ty, cz_ = cy;                                                            \
                bool more_z_ = false;                                                              \
                if (more_y) { BM_NEXT(ty, cy, tz_, cz_) more_z_ = tz_ < nb; }                      \
                { const int lt_ = more_z_ ? tz_ : ty, lc_ = more_z_ ? cz_ : cy; BMP_LOAD(lt_, lc_, UZ) } \
                BMP_GATHER(UY, GY)                                                                 \
                BMP_ADD(tx, cx, UX, GX)                                                            \
                if (!more_y) break;                                                                \
                if (ty != tx) BM_TOKEN_BARRIER;                                                    \
                tx = ty; cx = cy; more_y = more_z_;                                                \
                if (more_z_) { ty = tz_; cy = cz_; }        /* else (ty, cy) stays a valid (token, chunk) for the spare loads */ \
            }
            if (ti < nb) {
                int4u ua, ub, uc;
                double ga[4], gb[4], gc[4];
                int tx = ti, cx = c, ty = ti, cy = c;
                BMP_LOAD(tx, cx, ua)
                BM_NEXT(tx, cx, ty, cy)
                bool more_y = ty < nb;
                { const int lt_ = more_y ? ty : tx, lc_ = more_y ? cy : cx; BMP_LOAD(lt_, lc_, ub) }
                if (!more_y) { ty = tx; cy = cx; }
                BMP_GATHER(ua, ga)
                for (;;) {
                    BMP_TURN(ua, ga, ub, gb, uc)
                    BMP_TURN(ub, gb, uc, gc, ua)
                    BMP_TURN(uc, gc, ua, ga, ub)
                }
            }
#undef BMP_LOAD
#undef BMP_GATHER
#undef BMP_ADD
#undef BMP_TURN
#undef TOK_IDF
        } else
        if (ti < nb) {
            int4u da, db;
            double2u wa0, wa1, wb0, wb1;
#define BM_LOAD(TI, C, D, W0, W1)                                                                  \
            {                                                                                      \
                const int64_t o_ = TOK_A(TI) + min((C) * 4 * BM_THREADS + tid * 4, max(TOK_N(TI) - 1, 0)); \
                D = *reinterpret_cast<const int4u*>(doc + o_);                                     \
                W0 = *reinterpret_cast<const double2u*>(w + o_);                                   \
                W1 = *reinterpret_cast<const double2u*>(w + o_ + 2);                               \
            }
#define BM_ADD(TI, C, D, W0, W1)                                                                   \
            {                                                                                      \
                const int p_ = (C) * 4 * BM_THREADS + tid * 4, n_ = TOK_N(TI);                     \
                char* const sb_ = reinterpret_cast<char*>(sc);                                     \
                if (((C) + 1) * 4 * BM_THREADS <= n_) {      /* a full chunk (uniform): no tail tests */ \
                    double* a0_ = reinterpret_cast<double*>(sb_ + SC_OFF(D[0] - (int)base));       \
                    double* a1_ = reinterpret_cast<double*>(sb_ + SC_OFF(D[1] - (int)base));       \
                    double* a2_ = reinterpret_cast<double*>(sb_ + SC_OFF(D[2] - (int)base));       \
                    double* a3_ = reinterpret_cast<double*>(sb_ + SC_OFF(D[3] - (int)base));       \
                    const double v0_ = *a0_, v1_ = *a1_, v2_ = *a2_, v3_ = *a3_;                   \
                    *a0_ = v0_ + W0[0];                                                            \
                    *a1_ = v1_ + W0[1];                                                            \
                    *a2_ = v2_ + W1[0];                                                            \
                    *a3_ = v3_ + W1[1];                                                            \
                } else if (p_ < n_) {                                                              \
                    double* a0_ = reinterpret_cast<double*>(sb_ + SC_OFF(D[0] - (int)base));       \
                    double* a1_ = p_ + 1 < n_ ? reinterpret_cast<double*>(sb_ + SC_OFF(D[1] - (int)base)) : &sc[BM_SC_DOUBLES - 1]; \
                    double* a2_ = p_ + 2 < n_ ? reinterpret_cast<double*>(sb_ + SC_OFF(D[2] - (int)base)) : &sc[BM_SC_DOUBLES - 2]; \
                    double* a3_ = p_ + 3 < n_ ? reinterpret_cast<double*>(sb_ + SC_OFF(D[3] - (int)base)) : &sc[BM_SC_DOUBLES - 3]; \
                    const double v0_ = *a0_, v1_ = *a1_, v2_ = *a2_, v3_ = *a3_;                   \
                    *a0_ = v0_ + W0[0];                                                            \
                    if (p_ + 1 < n_) *a1_ = v1_ + W0[1];                                           \
                    if (p_ + 2 < n_) *a2_ = v2_ + W1[0];                                           \
                    if (p_ + 3 < n_) *a3_ = v3_ + W1[1];                                           \
                }                                                                                  \
            }
            BM_LOAD(ti, c, da, wa0, wa1)
            for (;;) {
                int nti, nc;
                BM_NEXT(ti, c, nti, nc)
                const bool more1 = nti < nb;
                { const int lt = more1 ? nti : ti, lc = more1 ? nc : c; BM_LOAD(lt, lc, db, wb0, wb1) }
                BM_ADD(ti, c, da, wa0, wa1)
                if (nti != ti) BM_TOKEN_BARRIER;
                if (!more1) break;
                ti = nti; c = nc;
                BM_NEXT(ti, c, nti, nc)
                const bool more2 = nti < nb;
                { const int lt = more2 ? nti : ti, lc = more2 ? nc : c; BM_LOAD(lt, lc, da, wa0, wa1) }
                BM_ADD(ti, c, db, wb0, wb1)
                if (nti != ti) BM_TOKEN_BARRIER;
                if (!more2) break;
                ti = nti; c = nc;
            }
#undef BM_LOAD
#undef BM_ADD
#undef BM_NEXT
#undef BM_TOKEN_BARRIER
#undef TOK_N
#undef TOK_A
        }
        __syncthreads();                                     // adds done before the next batch's metadata / the select
    }
    if (mode == 1) {
        double m = -INFINITY;
        for (int i = tid; i < lim; i += BM_THREADS) {
            const double v = sc[SC_IDX(i)];
            dense_out[(size_t)q * n_docs + base + i] = v;
            if (dx.raw32 != nullptr) dx.raw32[(size_t)q * dx.ld + base + i] = (float)v;
            if (dx.max_key != nullptr && (tenants == nullptr || tenants[base + i] == tenant)) m = fmax(m, v);
        }
        if (dx.max_key != nullptr) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
            if (lane == 0 && m > -INFINITY) atomicMax(&dx.max_key[q], (unsigned long long)f64_orderable(m));
        }
        return;
    }
    // ---- exact top-k of sc[0..lim): radix select of the k-th largest key, ties by lower doc ------------
    const size_t po = ((size_t)q * n_ranges + r) * k;
    // part_cnt[q][r] = number of valid entries at the front of this (query, range) slot group; the merge only reads those
    // Tenant filter (`WHERE agent_id = %s`, rag/document_store.py:457): a document of another tenant is an EMPTY slot
    // (key 0, below every real score) from here on, so it can neither enter a partial list nor move the threshold.
    if (lim <= k) {
        for (int i = tid; i < k; i += BM_THREADS) {
            const bool mine = i < lim && (tenants == nullptr || tenants[base + i] == tenant);
            part_key[po + i] = mine ? f64_orderable(sc[SC_IDX(i)]) : 0ull;
            part_row[po + i] = (uint32_t)(base + i);
        }
        if (tid == 0) part_cnt[(size_t)q * n_ranges + r] = lim;
        return;
    }
    const int seg0 = tid * BM_SEG;
    // this thread's BM_SEG docs, read once and UNCONDITIONALLY (the whole LDS array is initialised, a guarded read per doc was
    // eight serial LDS round trips); bit j of `valid` = doc seg0 + j exists and belongs to the tenant
    double sv[BM_SEG];
#pragma unroll
    for (int j = 0; j < BM_SEG; ++j) sv[j] = sc[SC_IDX(seg0 + j)];
    const int n_here = lim - seg0;
    unsigned valid = n_here >= BM_SEG ? (1u << BM_SEG) - 1u : (n_here <= 0 ? 0u : (1u << n_here) - 1u);
    if (tenants != nullptr) {
#pragma unroll
        for (int j = 0; j < BM_SEG; ++j)
            if ((valid >> j & 1u) && tenants[base + seg0 + j] != tenant) valid &= ~(1u << j);
    }
    // Thresholded ranges (second stage, see bm25_launch_topk): tau_key[q] is the k-th best key over the first-stage
    // ranges, a lower bound of the global k-th. Only keys >= tau can reach the global top-k; a later range holds
    // about k * BM_RANGE / (docs of stage one) of them, so they are compacted in doc order (the merge sorts anyway) and the
    // 8-pass radix select — 26 us per block, three quarters of this kernel's time when run for every range — is
    // skipped. More than k survivors (possible, e.g. a tie plateau) falls through to the exact select below.
    // The test runs on the scores themselves: key(s) > tau_key <=> s > the double tau_key stands for (the key map is a monotone
    // bijection of the non-NaN doubles; an accumulator is never -0.0: it starts at +0.0 and x + y is -0.0 only for two -0.0),
    // one compare per document instead of a key conversion and two 64-bit compares; only survivors are converted.
    if (tau_key != nullptr) {
        const uint64_t tb_ = (tk & 0x8000000000000000ull) ? (tk & 0x7fffffffffffffffull) : ~tk;
        const double tau_d = tk == 0ull ? -__builtin_inf() : __builtin_bit_cast(double, tb_);
        unsigned pass = 0;
#pragma unroll
        // STRICTLY above tau: the running list already holds k documents with score >= tau, all of them from earlier ranges =
        // lower document numbers, and ties go to the lower document - a document that only EQUALS tau can never displace one of
        // them. (Scores of short queries tie in droves - same tf, same length -; with >= a plateau at tau passed whole, which is
        // what pushed the second stage into the exact per-range select and its merge to 1.4-2.8 k entries per query.)
        for (int j = 0; j < BM_SEG; ++j) pass |= (sv[j] > tau_d ? 1u : 0u) << j;
        pass &= valid;
        const int n_in = __builtin_popcount(pass);
        int sc_in = n_in;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(sc_in, o);
            if (lane >= o) sc_in += up;
        }
        if (lane == 63) hist[wv] = sc_in;
        __syncthreads();
        int off = sc_in - n_in, total_in = 0;
        for (int u = 0; u < BM_THREADS / 64; ++u) {
            if (u < wv) off += hist[u];
            total_in += hist[u];
        }
        if (total_in <= k) {
            if (pass) {
#pragma unroll
                for (int j = 0; j < BM_SEG; ++j)
                    if (pass >> j & 1u) {
                        part_key[po + off] = f64_orderable(sv[j]);
                        part_row[po + off] = (uint32_t)(base + seg0 + j);
                        ++off;
                    }
            }
            if (tid == 0) part_cnt[(size_t)q * n_ranges + r] = total_in;
            return;
        }
        __syncthreads();                                   // (uniform: total_in is the same in every thread) hist is reused by the select below
    }
    uint64_t keys[BM_SEG];
#pragma unroll
    for (int j = 0; j < BM_SEG; ++j) keys[j] = (valid >> j & 1u) ? f64_orderable(sv[j]) : 0ull;
    uint64_t prefix = 0ull;          // matched high bytes of the pivot
    int remaining = k;               // the pivot is the `remaining`-th largest among keys matching `prefix`
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        int cur = -1, run = 0;
#pragma unroll
        for (int j = 0; j < BM_SEG; ++j) {
            const int i = seg0 + j;
            const uint64_t key = keys[j];
            const bool match = i < lim && (pass == 0 || (key >> (shift + 8)) == prefix);
            if (!match) continue;
            const int d = (int)((key >> shift) & 0xFF);
            if (d == cur) { ++run; } else {
                if (run) atomicAdd(&hist[cur], run);
                cur = d; run = 1;
            }
        }
        if (run) atomicAdd(&hist[cur], run);
        __syncthreads();
        // suffix sums over the 256 bins (4 waves x 64 bins), find the bin holding the `remaining`-th largest
        int suf = 0;
        if (tid < 256) {
            int v = hist[tid];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int up = __shfl_down(v, o);
                if (lane + o < 64) v += up;
            }
            suf = v;                                     // sum of bins [tid .. end of this wave's 64]
            if (lane == 0) wsum[wv] = v;
        }
        __syncthreads();
        if (tid < 256) {
            for (int u = wv + 1; u < 4; ++u) suf += wsum[u];      // bins of higher waves
            const int above = suf - hist[tid];                    // count of keys in bins > tid
            if (above < remaining && remaining <= suf) {          // exactly one bin satisfies this
                wsum[8] = tid;
                wsum[9] = remaining - above;
            }
        }
        __syncthreads();
        prefix = (prefix << 8) | (uint64_t)wsum[8];
        remaining = wsum[9];
        __syncthreads();
    }
    const uint64_t pivot = prefix;                      // exact key of the k-th largest
    const int need_ties = remaining;                    // how many keys == pivot belong to the top-k (lowest docs first)
    // ordered collection: every key > pivot, plus the first `need_ties` keys == pivot in doc order
    int n_gt = 0, n_eq = 0;
#pragma unroll
    for (int j = 0; j < BM_SEG; ++j) {
        const bool in = (seg0 + j) < lim;
        n_gt += in && keys[j] > pivot;
        n_eq += in && keys[j] == pivot;
    }
    // block exclusive scans of n_gt and n_eq (thread order == doc order)
    int s_gt = n_gt, s_eq = n_eq;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int a1 = __shfl_up(s_gt, o), a2 = __shfl_up(s_eq, o);
        if (lane >= o) { s_gt += a1; s_eq += a2; }
    }
    if (lane == 63) { hist[wv] = s_gt; hist[16 + wv] = s_eq; }
    __syncthreads();
    int off_gt = s_gt - n_gt, off_eq = s_eq - n_eq, total_gt = 0;
    for (int u = 0; u < BM_THREADS / 64; ++u) {
        if (u < wv) { off_gt += hist[u]; off_eq += hist[16 + u]; }
        total_gt += hist[u];
    }
#pragma unroll
    for (int j = 0; j < BM_SEG; ++j) {
        const int i = seg0 + j;
        if (i >= lim) continue;
        const uint64_t key = keys[j];
        if (key > pivot) {
            part_key[po + off_gt] = key;
            part_row[po + off_gt] = (uint32_t)(base + i);
            ++off_gt;
        } else if (key == pivot) {
            if (off_eq < need_ties) {
                part_key[po + total_gt + off_eq] = key;
                part_row[po + total_gt + off_eq] = (uint32_t)(base + i);
            }
            ++off_eq;
        }
    }
    if (tid == 0) part_cnt[(size_t)q * n_ranges + r] = k;
}

#define BM_MERGE 2048
__device__ __forceinline__ void bm_sort_pairs(uint64_t* k1, uint32_t* k2, int P, int tid, int nthreads) {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P; i += nthreads) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const bool first_block = ((i & k) == 0);
                    const bool a_before_b = k1[i] > k1[ixj] || (k1[i] == k1[ixj] && k2[i] < k2[ixj]);
                    if (first_block ? !a_before_b : a_before_b) {
                        const uint64_t t1 = k1[i]; k1[i] = k1[ixj]; k1[ixj] = t1;
                        const uint32_t t2 = k2[i]; k2[i] = k2[ixj]; k2[ixj] = t2;
                    }
                }
            }
            __syncthreads();
        }
    }
}

#define BM_FIRST_RANGES 4      // exact per-range top-k for these (8192 docs), thresholded compaction for the rest (2048-doc ranges: 2 -> 134.2 k hybrid q/s, 4 -> 151.2 k;
                               // 4096-doc ranges: 2 -> 144.8 k, 4 -> 143.3 k; 16384-doc ranges, BM25 leg alone: 1 -> 7.5 ms, 2 -> 4.87, 4 -> 5.08)
#ifndef BM_STAGE_GROWTH        // -DBM_STAGE_GROWTH: variant builds (tools/lib_ab.sh; r3: 4 -> 3.04 ms, 8 -> 2.88, 16 -> 3.05, 32 -> 3.44 per 1024-query batch)
#define BM_STAGE_GROWTH 8      // every thresholded stage covers up to 8x the ranges seen before it
#endif

// Folds the partial lists of the doc ranges [r_begin, r_end) into the query's RUNNING top-k (run_key / run_row [Q][k], key 0 =
// empty) and publishes tau_key[q] = its k-th key: a lower bound of the global k-th best, 0 ("everything passes") while fewer
// than k documents have been seen. One workgroup per query; windows of BM_MERGE - k new entries, bitonic sort in LDS.
// A thresholded range fills only the first part_cnt[q][r] of its k slots and the rest of the slot group holds whatever an
// earlier batch left there, so ONLY the counted fronts are read, for any number of ranges: they are walked in groups of 256
// (one prefix scan per group; a 12.5M-document shard has 763 ranges).
// last != 0: the running list is final -> ids / rows / scores (divided by max if > 0 else 1.0, rag/retrieval.py:343-345).
__global__ __launch_bounds__(256) void bm25_merge_stage_kernel(const uint64_t* __restrict__ part_key, const uint32_t* __restrict__ part_row,
                                                                const int* __restrict__ part_cnt, int n_ranges, int r_begin, int r_end,
                                                                int k, uint64_t* __restrict__ run_key, uint32_t* __restrict__ run_row,
                                                                int first, uint64_t* __restrict__ tau_key, int last,
                                                                const int64_t* __restrict__ idmap, int64_t id_base,
                                                                int64_t* __restrict__ ids_out, int32_t* __restrict__ rows_out,
                                                                double* __restrict__ scores_out, double* __restrict__ raw_max_out,
                                                                int normalize) {
    __shared__ uint64_t sk[BM_MERGE];
    __shared__ uint32_t sr[BM_MERGE];
    __shared__ int pre[257];                        // exclusive prefix of the entry counts of the current group of <= 256 ranges
    __shared__ int wtot[4];
    const int q = blockIdx.x, tid = threadIdx.x;
    const uint64_t* pk = part_key + (size_t)q * n_ranges * k;
    const uint32_t* pr = part_row + (size_t)q * n_ranges * k;
    for (int i = tid; i < BM_MERGE; i += 256) {
        const bool keep = !first && i < k;
        sk[i] = keep ? run_key[(size_t)q * k + i] : 0ull;
        sr[i] = keep ? run_row[(size_t)q * k + i] : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (int r0 = r_begin; r0 < r_end; r0 += 256) {
        const int nr = min(256, r_end - r0);
        const int c = tid < nr ? min(part_cnt[(size_t)q * n_ranges + r0 + tid], k) : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += up;
        }
        if ((tid & 63) == 63) wtot[tid >> 6] = incl;
        __syncthreads();
        int off = 0;
        for (int w = 0; w < (tid >> 6); ++w) off += wtot[w];
        pre[tid + 1] = off + incl;
        if (tid == 0) pre[0] = 0;
        __syncthreads();
        const int total = pre[nr];
        int pos = 0;
        while (pos < total) {
            // window = the smallest power of two that holds the running top-k plus what is left of this group (a stage usually
            // brings a few hundred entries: sorting 512 or 1024 slots instead of 2048 halves the merge time)
            int P = 256;
            while (P < BM_MERGE && P < k + (total - pos)) P <<= 1;
            const int room = P - k;
            const int take = min(room, total - pos);
            for (int i = tid; i < room; i += 256) {
                uint64_t key = 0ull;
                uint32_t row = 0xFFFFFFFFu;
                if (i < take) {                     // entry e -> range r with pre[r] <= e < pre[r+1] -> slot r*k + (e - pre[r])
                    const int e = pos + i;
                    int lo = 0, hi = nr;
                    while (hi - lo > 1) {
                        const int mid = (lo + hi) >> 1;
                        if (pre[mid] <= e) lo = mid; else hi = mid;
                    }
                    const size_t slot = (size_t)(r0 + lo) * k + (e - pre[lo]);
                    key = pk[slot];
                    row = pr[slot];
                }
                sk[k + i] = key;
                sr[k + i] = row;
            }
            __syncthreads();
            bm_sort_pairs(sk, sr, P, tid, 256);
            pos += take;
        }
        __syncthreads();                            // every thread is done with pre[] / wtot[] before the next group's scan
    }
    for (int i = tid; i < k; i += 256) {
        run_key[(size_t)q * k + i] = sk[i];
        run_row[(size_t)q * k + i] = sr[i];
    }
    if (tid == 0 && tau_key != nullptr) tau_key[q] = sk[k - 1];        // 0 if fewer than k real keys so far
    if (!last) return;
    // top-1 raw score -> divisor (max if > 0 else 1.0), retrieval.py:344
    uint64_t u0 = sk[0];
    double mx = 1.0;
    if (u0 != 0ull) {
        u0 = (u0 & 0x8000000000000000ull) ? (u0 & 0x7fffffffffffffffull) : ~u0;
        const double top = __builtin_bit_cast(double, u0);
        if (top > 0.0) mx = top;
    }
    if (tid == 0 && raw_max_out) raw_max_out[q] = mx;
    for (int i = tid; i < k; i += 256) {
        const bool ok = sk[i] != 0ull;
        uint64_t u = sk[i];
        u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
        const double s = __builtin_bit_cast(double, u);
        ids_out[(size_t)q * k + i] = ok ? (idmap ? idmap[sr[i]] : id_base + (int64_t)sr[i]) : -1;
        if (rows_out) rows_out[(size_t)q * k + i] = ok ? (int32_t)sr[i] : -1;
        scores_out[(size_t)q * k + i] = ok ? (normalize ? s / mx : s) : 0.0;
    }
}

// ---- the same fold by SELECTION instead of sorting (k <= BMS_KMAX; the default) ----------------------------------------------
// The bitonic merge above sorts 512 .. 2048 slots per query and stage to keep 100 of them: 42 / 225 / 121 / 21 us for the four
// stages of a 1024-query batch at 1M documents, 0.41 of the 2.9 ms BM25 leg. Here ONE WAVE folds one query: the running list
// (sorted, k slots at the front of the wave's LDS window) and the stage's new entries (appended behind it, straight from the
// counted fronts of the partial lists) are loaded into registers, the k-th largest key is found bit by bit with ballots (64
// steps x one ballot per key register, no atomics, no sort - the dense path's select), ties at the k-th key are cut by the
// LOWEST ROWS with a second bitwise search over the rows of the tied entries, the k survivors are compacted and ranked by
// counting (k x k / 64 compares per lane). Result: the same list bit for bit (order: key descending, row ascending).
#define BMS_KMAX 256
#define BMS_CAP 2048                     // entries a wave selects from at once: k running + up to BMS_CAP - k new ones
#define BMS_WAVES 2
#define BMS_WAVE_BYTES (BMS_CAP * 12 + BMS_KMAX * 12)
#define BMS_LDS_BYTES (BMS_WAVES * BMS_WAVE_BYTES)
#define BMS_FENCE asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")       // one wave: LDS operations retire in order; this keeps the compiler honest

// top-k of wk/wr[0, total) -> wk/wr[0, k) sorted (key descending, row ascending), empty slots (0, 0xFFFFFFFF) behind
template <int NREG>
__device__ __forceinline__ void bms_select(uint64_t* __restrict__ wk, uint32_t* __restrict__ wr, uint64_t* __restrict__ tk,
                                           uint32_t* __restrict__ tr, int total, int k, int lane) {
    uint64_t key[NREG];
    uint32_t row[NREG];
#pragma unroll
    for (int e = 0; e < NREG; ++e) {
        const int i = e * 64 + lane;
        key[e] = i < total ? wk[i] : 0ull;
        row[e] = i < total ? wr[i] : 0xFFFFFFFFu;
    }
    BMS_FENCE;
    int n_valid = 0;
#pragma unroll
    for (int e = 0; e < NREG; ++e) n_valid += __popcll(__ballot(key[e] != 0ull));
    uint64_t pivot = 0ull;               // survivors: key > pivot, or key == pivot and ~row >= inv_pivot (0, 0: every real key)
    uint32_t inv_pivot = 0u;
    if (n_valid > k) {
        pivot = 0ull;
        int ge = 0;
        for (int bit = 63; bit >= 0; --bit) {
            const uint64_t trial = pivot | (1ull << bit);
            int c = 0;
#pragma unroll
            for (int e = 0; e < NREG; ++e) c += __popcll(__ballot(key[e] >= trial));
            if (c >= k) { pivot = trial; ge = c; }
            if (c == k) break;
        }
        if (ge > k) {                    // a plateau at the k-th key: the m tied entries with the lowest rows stay
            int gt = 0;
#pragma unroll
            for (int e = 0; e < NREG; ++e) gt += __popcll(__ballot(key[e] > pivot));
            const int m = k - gt;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t trial = inv_pivot | (1u << bit);
                int c = 0;
#pragma unroll
                for (int e = 0; e < NREG; ++e) c += __popcll(__ballot(key[e] == pivot && ~row[e] >= trial));
                if (c >= m) inv_pivot = trial;
                if (c == m) break;
            }
        }
    }
    // compaction (any order) into the scratch list
    int n_s = 0;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int e = 0; e < NREG; ++e) {
        const bool in = key[e] != 0ull && (key[e] > pivot || (key[e] == pivot && ~row[e] >= inv_pivot));
        const uint64_t bt = __ballot(in);
        if (in) {
            const int d = n_s + __popcll(bt & lt_mask);
            tk[d] = key[e];
            tr[d] = row[e];
        }
        n_s += __popcll(bt);
    }
    BMS_FENCE;
    // rank by counting: slot of a survivor = the number of survivors that come before it
    for (int i = lane; i < k; i += 64) {
        wk[i] = 0ull;
        wr[i] = 0xFFFFFFFFu;
    }
    BMS_FENCE;
    for (int i = lane; i < n_s; i += 64) {
        const uint64_t a = tk[i];
        const uint32_t ar = tr[i];
        int rank = 0;
        for (int j = 0; j < n_s; ++j) {
            const uint64_t b = tk[j];
            rank += (b > a) || (b == a && tr[j] < ar);
        }
        wk[rank] = a;
        wr[rank] = ar;
    }
    BMS_FENCE;
}

__global__ __launch_bounds__(64 * BMS_WAVES) void bm25_merge_select_kernel(const uint64_t* __restrict__ part_key, const uint32_t* __restrict__ part_row,
                                                                         const int* __restrict__ part_cnt, int n_ranges, int r_begin, int r_end,
                                                                         int k, uint64_t* __restrict__ run_key, uint32_t* __restrict__ run_row,
                                                                         int first, uint64_t* __restrict__ tau_key, int last,
                                                                         const int64_t* __restrict__ idmap, int64_t id_base,
                                                                         int64_t* __restrict__ ids_out, int32_t* __restrict__ rows_out,
                                                                         double* __restrict__ scores_out, double* __restrict__ raw_max_out,
                                                                         int normalize, int n_queries) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int q = blockIdx.x * BMS_WAVES + wv;
    if (q >= n_queries) return;                              // whole wave; nothing below synchronises across waves
    uint64_t* wk = reinterpret_cast<uint64_t*>(smem + (size_t)wv * BMS_WAVE_BYTES);      // [BMS_CAP]
    uint64_t* tk = wk + BMS_CAP;                                                         // [BMS_KMAX]
    uint32_t* wr = reinterpret_cast<uint32_t*>(tk + BMS_KMAX);                           // [BMS_CAP]
    uint32_t* tr = wr + BMS_CAP;                                                         // [BMS_KMAX]
    const uint64_t* pk = part_key + (size_t)q * n_ranges * k;
    const uint32_t* pr = part_row + (size_t)q * n_ranges * k;
    for (int i = lane; i < k; i += 64) {
        wk[i] = first ? 0ull : run_key[(size_t)q * k + i];
        wr[i] = first ? 0xFFFFFFFFu : run_row[(size_t)q * k + i];
    }
    int n_win = 0;                                           // new entries behind the running list
#define BMS_FLUSH                                                                      \
    {                                                                                  \
        BMS_FENCE;                                                                     \
        const int tot_ = k + n_win;                                                    \
        if (tot_ <= 512) bms_select<8>(wk, wr, tk, tr, tot_, k, lane);                 \
        else if (tot_ <= 1024) bms_select<16>(wk, wr, tk, tr, tot_, k, lane);          \
        else bms_select<32>(wk, wr, tk, tr, tot_, k, lane);                            \
        n_win = 0;                                                                     \
    }
    for (int r0 = r_begin; r0 < r_end; r0 += 64) {
        const int nr = min(64, r_end - r0);
        const int c = lane < nr ? min(part_cnt[(size_t)q * n_ranges + r0 + lane], k) : 0;
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        int done = 0, base = 0;                              // ranges of this group already taken, and their entries
        while (done < nr) {
            const int room = BMS_CAP - k - n_win;
            const bool acc = lane >= done && lane < nr && incl - base <= room;
            const int n_acc = __popcll(__ballot(acc));       // a prefix of the remaining ranges (incl is monotone)
            if (n_acc > 0) {
                const int off = k + n_win + (incl - c - base);
                const bool small = acc && c <= 4;
                if (small) {
                    const size_t slot = (size_t)(r0 + lane) * k;
                    for (int j = 0; j < c; ++j) {
                        wk[off + j] = pk[slot + j];
                        wr[off + j] = pr[slot + j];
                    }
                }
                uint64_t big = __ballot(acc && c > 4);       // longer fronts (an exact range brings k entries): the whole wave copies
                while (big) {
                    const int L = __ffsll((unsigned long long)big) - 1;
                    big &= big - 1;
                    const int cL = __builtin_amdgcn_readlane(c, L), offL = __builtin_amdgcn_readlane(off, L);
                    const size_t slot = (size_t)(r0 + L) * k;
                    for (int j = lane; j < cL; j += 64) {
                        wk[offL + j] = pk[slot + j];
                        wr[offL + j] = pr[slot + j];
                    }
                }
                const int upto = __builtin_amdgcn_readlane(incl, done + n_acc - 1);
                n_win += upto - base;
                base = upto;
                done += n_acc;
            }
            if (done < nr) BMS_FLUSH                         // the window is full: fold, then go on with the rest of the group
        }
    }
    if (n_win > 0 || first) BMS_FLUSH
#undef BMS_FLUSH
    BMS_FENCE;
    for (int i = lane; i < k; i += 64) {
        run_key[(size_t)q * k + i] = wk[i];
        run_row[(size_t)q * k + i] = wr[i];
    }
    if (lane == 0 && tau_key != nullptr) tau_key[q] = wk[k - 1];      // 0 if fewer than k real keys so far
    if (!last) return;
    uint64_t u0 = wk[0];
    double mx = 1.0;
    if (u0 != 0ull) {
        u0 = (u0 & 0x8000000000000000ull) ? (u0 & 0x7fffffffffffffffull) : ~u0;
        const double top = __builtin_bit_cast(double, u0);
        if (top > 0.0) mx = top;
    }
    if (lane == 0 && raw_max_out) raw_max_out[q] = mx;
    for (int i = lane; i < k; i += 64) {
        const bool ok = wk[i] != 0ull;
        uint64_t u = wk[i];
        u = (u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u;
        const double sc_ = __builtin_bit_cast(double, u);
        ids_out[(size_t)q * k + i] = ok ? (idmap ? idmap[wr[i]] : id_base + (int64_t)wr[i]) : -1;
        if (rows_out) rows_out[(size_t)q * k + i] = ok ? (int32_t)wr[i] : -1;
        scores_out[(size_t)q * k + i] = ok ? (normalize ? sc_ / mx : sc_) : 0.0;
    }
}

// per-call device buffers of a top-k search: partial lists [Q][n_ranges][k], counts [Q][n_ranges], running list, threshold
// + the per-call plan (see bm25_plan_kernel): plan.off [Q][BM_PLAN_T][n_ranges + 1] int32, plan.meta [Q][BM_PLAN_T]
struct bm25_plan_ws { int32_t* off; bm_plan_meta* meta; };
struct bm25_topk_ws {
    uint64_t* part_key; uint32_t* part_row; int* part_cnt; uint64_t* run_key; uint32_t* run_row; uint64_t* tau;
    bm25_plan_ws plan;
};
struct bm25_topk_out {
    const int64_t* idmap; int64_t id_base; int64_t* ids; int32_t* rows; double* scores; double* raw_max; int normalize;
};

// STAGED top-k over all doc ranges: ranges [0, first) get the exact per-range top-k; after every stage the running top-k
// of the query gives tau (k-th best so far), and the next stage - up to BM_STAGE_GROWTH x the ranges seen so far - only
// compacts keys >= tau. Expected survivors per stage ~ k * growth per query however large the shard is (a single
// threshold from 32768 docs left ~k/2 per range: 38 k entries per query to merge on a 12.5M-doc shard).
// the scoring launch: packed 4-byte postings when the index has them, the 12-byte form otherwise
#define BM_RANGE_LAUNCH(H, IX, NR_L, NQ, ST, DX, NR, TP, TM, K, MODE, ...)                                                             \
    {                                                                                                                              \
        bm_grid gm_ = bm_make_grid(NR_L, NQ, (H)->opt.bm25_linear_grid);                                                           \
        gm_.plan_t = (IX)->plan_t;                                                                                                 \
        if ((IX)->packed != nullptr)                                                                                               \
            hipLaunchKernelGGL(bm25_range_kernel<true>, dim3(gm_.blocks), dim3(BM_THREADS), BM_LDS_BYTES, ST, (IX)->meta, (IX)->doc, (IX)->w, \
                               (IX)->packed, (IX)->gtab, (IX)->range_tab, NR, TP, TM, (IX)->n_docs, (IX)->n_terms, K, MODE, __VA_ARGS__, gm_, DX); \
        else                                                                                                                       \
            hipLaunchKernelGGL(bm25_range_kernel<false>, dim3(gm_.blocks), dim3(BM_THREADS), BM_LDS_BYTES, ST, (IX)->meta, (IX)->doc, (IX)->w, \
                               (IX)->packed, (IX)->gtab, (IX)->range_tab, NR, TP, TM, (IX)->n_docs, (IX)->n_terms, K, MODE, __VA_ARGS__, gm_, DX); \
    }
static int bm25_pick_plan_t(const rag_ctx* h, const rag_bm25_index* ix, int Q) {
    if (h->opt.bm25_plan_slots > 0) return std::min(BM_PLAN_T, h->opt.bm25_plan_slots);
    const size_t per_slot = (size_t)Q * (ix->n_ranges + 1) * sizeof(int32_t);
    return (int)std::min<size_t>(BM_PLAN_T, std::max<size_t>(8, BM_PLAN_BUDGET / std::max<size_t>(1, per_slot)));
}
static size_t bm25_plan_off_entries(const rag_bm25_index* ix, int Q) { return (size_t)Q * ix->plan_t * (ix->n_ranges + 1); }
static void bm25_launch_plan(const rag_bm25_index* ix, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, const bm25_plan_ws& p,
                             hipStream_t st) {
    hipLaunchKernelGGL(bm25_plan_kernel, dim3((ix->n_ranges + 1 + 255) / 256, Q), dim3(256), 0, st, ix->meta, ix->doc, ix->range_tab,
                       ix->n_ranges, ix->n_terms, term_ptr_dev, terms_dev, p.off, p.meta, ix->plan_t);
}

static void bm25_launch_topk(const rag_ctx* h, const rag_bm25_index* ix, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k,
                             const bm25_topk_ws& w, const bm25_topk_out& o, const int32_t* tenants, int tenant, hipStream_t st) {
    const int nr = ix->n_ranges;
    bm25_launch_plan(ix, term_ptr_dev, terms_dev, Q, w.plan, st);
    const int first_cfg = h->opt.bm25_first_ranges >= 1 && h->opt.bm25_first_ranges <= 16 ? h->opt.bm25_first_ranges : BM_FIRST_RANGES;
    const bool staged = nr > 2 * first_cfg && !h->opt.bm25_no_staging;
    int begin = 0, stage = 0;
    while (begin < nr) {
        int end = !staged ? nr : (stage == 0 ? first_cfg : (int)std::min<int64_t>(nr, (int64_t)begin * BM_STAGE_GROWTH));
        if (staged && nr - end < end / 4) end = nr;                   // no tiny trailing stage
        BM_RANGE_LAUNCH(h, ix, end - begin, Q, st, (bm_dense_extra{nullptr, 0, nullptr}), nr, term_ptr_dev, terms_dev, k, 0, (double*)nullptr,
                        w.part_key, w.part_row, begin, stage == 0 ? (const uint64_t*)nullptr : (const uint64_t*)w.tau, w.part_cnt,
                        tenants, tenant, (const int32_t*)w.plan.off, (const bm_plan_meta*)w.plan.meta)
        const int last = end == nr;
        if (k <= BMS_KMAX && !h->opt.bm25_sort_merge)
            hipLaunchKernelGGL(bm25_merge_select_kernel, dim3((Q + BMS_WAVES - 1) / BMS_WAVES), dim3(64 * BMS_WAVES), BMS_LDS_BYTES, st,
                               w.part_key, w.part_row, w.part_cnt, nr, begin, end, k, w.run_key, w.run_row, stage == 0 ? 1 : 0, w.tau, last,
                               o.idmap, o.id_base, o.ids, o.rows, o.scores, o.raw_max, o.normalize, Q);
        else
            hipLaunchKernelGGL(bm25_merge_stage_kernel, dim3(Q), dim3(256), 0, st, w.part_key, w.part_row, w.part_cnt, nr, begin, end, k,
                               w.run_key, w.run_row, stage == 0 ? 1 : 0, w.tau, last, o.idmap, o.id_base, o.ids, o.rows, o.scores,
                               o.raw_max, o.normalize);
        begin = end;
        ++stage;
    }
}

// ------------------------------------------------------------------------------------------------
static void bm25_index_free(rag_bm25_index* ix) {
    if (!ix) return;
    hipFree(ix->indptr); hipFree(ix->doc); hipFree(ix->w); hipFree(ix->idf); hipFree(ix->meta); hipFree(ix->range_tab);
    hipFree(ix->packed); hipFree(ix->gtab);
    hipFree(ix->ws_key); hipFree(ix->ws_row); hipFree(ix->ws_tau); hipFree(ix->ws_cnt); hipFree(ix->ws_run_key); hipFree(ix->ws_run_row);
    hipFree(ix->ws_plan_off); hipFree(ix->ws_plan_meta);
    delete ix;
}

void bm25_free(rag_ctx* h) {
    bm25_index_free(h->bm25);
    h->bm25 = nullptr;
}

// host CSR -> device index (impacts + range table). Synchronous. On failure *out is freed and left null.
static int bm25_build(rag_ctx* h, const int64_t* indptr, const int32_t* doc, const int32_t* tf, const int32_t* doc_len,
                      const double* idf, int64_t n_docs, int64_t n_terms, double avgdl, double k1, double b, rag_bm25_index** out) {
    *out = nullptr;
    ARG_CHECK(h, n_docs > 0 && n_terms >= 0 && n_docs < 0x7fffffff, "bm25_load: bad sizes");
    ARG_CHECK(h, indptr && doc_len && (n_terms == 0 || idf), "bm25_load: null pointer");
    const int64_t nnz = n_terms ? indptr[n_terms] : 0;
    ARG_CHECK(h, nnz == 0 || (doc && tf), "bm25_load: null postings");
    rag_bm25_index* ix = new rag_bm25_index();
    ix->n_docs = n_docs; ix->n_terms = n_terms; ix->nnz = nnz; ix->avgdl = avgdl; ix->k1 = k1; ix->b = b;
    hipStream_t st = h->stream;
    int32_t *tfd = nullptr, *dld = nullptr;
    ix->n_ranges = (int)((n_docs + BM_RANGE - 1) / BM_RANGE);
    // per-term metadata + the plan of the bracket tables (host: one pass over indptr)
    const int64_t n_pad = (int64_t)ix->n_ranges * BM_RANGE;
    std::vector<bm_term_meta> meta_h((size_t)std::max<int64_t>(1, n_terms));
    int64_t n_tab = 0;
    for (int64_t t = 0; t < n_terms; ++t) {
        const int64_t df = indptr[t + 1] - indptr[t];
        if (df < 0 || df > n_docs) {
            delete ix;
            h->err = "bad argument: bm25_load: indptr must be non-decreasing with at most n_docs postings per term";
            return RAG_ERR_ARG;
        }
        int64_t e_t = 0;
        const int g = bm_plan_term(df, n_pad, &e_t);
        meta_h[(size_t)t] = {indptr[t], n_tab, idf[t], (int32_t)df, g};
        n_tab += e_t;
    }
    ix->tab_entries = n_tab;
    // code space of the packed postings: distinct term frequencies x distinct document lengths (host scans of tf[] and doc_len[])
    const int64_t code_cap = (int64_t)1 << (32 - BM_RANGE_LOG2);
    std::vector<uint32_t> tf_rank_h, dl_rank_of_doc_h;
    std::vector<double> tf_values_h, dl_values_h;
    bool packed_ok = nnz > 0 && h->opt.bm25_packed;      // opt-in: a third less HBM for the postings, a slower scoring loop
    if (packed_ok) {
        int32_t tf_max = 0, dl_max = 0;
        bool bad = false;
        for (int64_t p = 0; p < nnz; ++p) { const int32_t v = tf[p]; bad |= v <= 0; tf_max = v > tf_max ? v : tf_max; }
        for (int64_t d = 0; d < n_docs; ++d) { const int32_t v = doc_len[d]; bad |= v < 0; dl_max = v > dl_max ? v : dl_max; }
        packed_ok = !bad && tf_max < (1 << 24) && dl_max < (1 << 26);
        if (packed_ok) {
            std::vector<char> seen_tf((size_t)tf_max + 1, 0), seen_dl((size_t)dl_max + 1, 0);
            for (int64_t p = 0; p < nnz; ++p) seen_tf[(size_t)tf[p]] = 1;
            for (int64_t d = 0; d < n_docs; ++d) seen_dl[(size_t)doc_len[d]] = 1;
            tf_rank_h.assign((size_t)tf_max + 1, 0);
            std::vector<uint32_t> dl_rank((size_t)dl_max + 1, 0);
            for (int32_t v = 0; v <= tf_max; ++v) if (seen_tf[(size_t)v]) { tf_rank_h[(size_t)v] = (uint32_t)tf_values_h.size(); tf_values_h.push_back((double)v); }
            for (int32_t v = 0; v <= dl_max; ++v) if (seen_dl[(size_t)v]) { dl_rank[(size_t)v] = (uint32_t)dl_values_h.size(); dl_values_h.push_back((double)v); }
            ix->n_codes = (int64_t)tf_values_h.size() * (int64_t)dl_values_h.size();
            packed_ok = ix->n_codes <= code_cap;
            if (packed_ok) {
                dl_rank_of_doc_h.resize((size_t)n_docs);
                for (int64_t d = 0; d < n_docs; ++d) dl_rank_of_doc_h[(size_t)d] = dl_rank[(size_t)doc_len[d]];
            }
        }
    }
    if (!packed_ok) ix->n_codes = 0;
    // + 8 postings of padding: the scoring kernel reads 4 consecutive postings per thread, the bracket search 8 doc ids, without
    // a bounds branch
    uint32_t *tfr_d = nullptr, *dlr_d = nullptr;
    double *tfv_d = nullptr, *dlv_d = nullptr;
    hipError_t e = hipMalloc(&ix->indptr, (size_t)(n_terms + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&ix->doc, (size_t)(nnz + 8) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemsetAsync(ix->doc + nnz, 0, 8 * sizeof(int32_t), st);
    if (packed_ok) {
        if (e == hipSuccess) e = hipMalloc(&ix->packed, (size_t)(nnz + 8) * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMemsetAsync(ix->packed + nnz, 0, 8 * sizeof(uint32_t), st);
        if (e == hipSuccess) e = hipMalloc(&ix->gtab, (size_t)ix->n_codes * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&tfr_d, tf_rank_h.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&dlr_d, dl_rank_of_doc_h.size() * sizeof(uint32_t));
        if (e == hipSuccess) e = hipMalloc(&tfv_d, tf_values_h.size() * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&dlv_d, dl_values_h.size() * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(tfr_d, tf_rank_h.data(), tf_rank_h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dlr_d, dl_rank_of_doc_h.data(), dl_rank_of_doc_h.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(tfv_d, tf_values_h.data(), tf_values_h.size() * sizeof(double), hipMemcpyHostToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(dlv_d, dl_values_h.data(), dl_values_h.size() * sizeof(double), hipMemcpyHostToDevice, st);
    } else {
        if (e == hipSuccess) e = hipMalloc(&ix->w, (size_t)(nnz + 8) * sizeof(double));
        if (e == hipSuccess) e = hipMemsetAsync(ix->w + nnz, 0, 8 * sizeof(double), st);
    }
    if (e == hipSuccess) e = hipMalloc(&ix->idf, std::max<size_t>(1, n_terms) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&ix->meta, meta_h.size() * sizeof(bm_term_meta));
    if (e == hipSuccess) e = hipMalloc(&tfd, std::max<size_t>(1, nnz) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&dld, (size_t)n_docs * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc(&ix->range_tab, std::max<size_t>(1, (size_t)n_tab) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpyAsync(ix->indptr, indptr, (size_t)(n_terms + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(ix->doc, doc, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nnz) e = hipMemcpyAsync(tfd, tf, (size_t)nnz * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_terms) e = hipMemcpyAsync(ix->idf, idf, (size_t)n_terms * sizeof(double), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && n_terms) e = hipMemcpyAsync(ix->meta, meta_h.data(), (size_t)n_terms * sizeof(bm_term_meta), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(dld, doc_len, (size_t)n_docs * sizeof(int32_t), hipMemcpyHostToDevice, st);
    if (e == hipSuccess && nnz && !packed_ok) {
        hipLaunchKernelGGL(bm25_weights_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, ix->indptr, ix->doc, tfd,
                           dld, nnz, avgdl, k1, b, ix->idf, n_terms, ix->w);
        e = hipGetLastError();
    }
    if (e == hipSuccess && packed_ok) {
        hipLaunchKernelGGL(bm25_pack_kernel, dim3((unsigned)((nnz + 255) / 256)), dim3(256), 0, st, ix->doc, tfd, tfr_d, dlr_d, nnz,
                           (uint32_t)dl_values_h.size(), ix->packed);
        hipLaunchKernelGGL(bm25_gtab_kernel, dim3((unsigned)((ix->n_codes + 255) / 256)), dim3(256), 0, st, tfv_d, dlv_d, ix->n_codes,
                           (uint32_t)dl_values_h.size(), avgdl, k1, b, ix->gtab);
        e = hipGetLastError();
    }
    if (e == hipSuccess && n_tab) {
        hipLaunchKernelGGL(bm25_range_table_kernel, dim3((unsigned)((n_tab + 255) / 256)), dim3(256), 0, st, ix->meta, ix->doc,
                           n_terms, n_tab, ix->range_tab);
        e = hipGetLastError();
    }
    const hipError_t e2 = hipStreamSynchronize(st);               // (also keeps meta_h alive until its copy has been read)
    hipFree(tfd); hipFree(dld); hipFree(tfr_d); hipFree(dlr_d); hipFree(tfv_d); hipFree(dlv_d);
    hipFree(ix->indptr); hipFree(ix->idf);                         // only the builders above read them: the metadata carries both
    ix->indptr = nullptr; ix->idf = nullptr;
    if (e != hipSuccess || e2 != hipSuccess) {
        bm25_index_free(ix);
        h->err = std::string("bm25_load: ") + hipGetErrorString(e != hipSuccess ? e : e2);
        return RAG_ERR_HIP;
    }
    *out = ix;
    return RAG_OK;
}

int bm25_load_host(rag_ctx* h, const int64_t* indptr, const int32_t* doc, const int32_t* tf, const int32_t* doc_len,
                   const double* idf, int64_t n_docs, int64_t n_terms, double avgdl, double k1, double b) {
    rag_bm25_index* ix = nullptr;
    const int rc = bm25_build(h, indptr, doc, tf, doc_len, idf, n_docs, n_terms, avgdl, k1, b, &ix);
    if (rc) return rc;
    bm25_free(h);
    h->bm25 = ix;
    return RAG_OK;
}

static int bm25_set_attr(rag_ctx* h) {
    if (!h->attr_bm25) {
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bm25_range_kernel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, BM_LDS_BYTES));
        HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(bm25_range_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, BM_LDS_BYTES));
        h->attr_bm25 = true;
    }
    return RAG_OK;
}

static int bm25_tenant_args(rag_ctx* h, const rag_bm25_index* ix, int tenant, const int32_t** tenants_out) {
    *tenants_out = nullptr;
    if (tenant < 0) return RAG_OK;
    ARG_CHECK(h, h->tenants != nullptr && h->n_rows == ix->n_docs,
              "bm25: a tenant filter needs rag_index_set_tenants_host and postings row-aligned with the index");
    *tenants_out = h->tenants;
    return RAG_OK;
}

// host-pointer search against `ix` (the resident index or an ad-hoc one). Device staging comes from the handle's
// grow-only arena: no hipMalloc / hipFree per call.
static int bm25_run(rag_ctx* h, rag_bm25_index* ix, const int32_t* term_ptr, const int32_t* terms, int Q, int k, int mode, int tenant,
                    int64_t* ids_out, int32_t* rows_out, double* scores_out, double* raw_max_out, double* dense_out) {
    ARG_CHECK(h, ix != nullptr, "no BM25 index loaded");
    ARG_CHECK(h, Q > 0 && Q <= 65535 && term_ptr, "bm25: 1 <= n_queries <= 65535");
    const int n_terms_q = term_ptr[Q];
    ARG_CHECK(h, n_terms_q >= 0 && (n_terms_q == 0 || terms), "bm25: bad term arrays");
    if (mode == 0) ARG_CHECK(h, k > 0 && k <= BM_MERGE / 2 && k <= BM_RANGE, "bm25: 0 < k <= 1024");
    const int32_t* tenants = nullptr;
    int rc = bm25_tenant_args(h, ix, mode == 0 ? tenant : -1, &tenants);
    if (rc) return rc;
    if ((rc = bm25_set_attr(h))) return rc;
    hipStream_t st = h->stream;
    const int nr = ix->n_ranges;
    const size_t n_part = mode == 0 ? (size_t)Q * nr * k : 0, n_out = mode == 0 ? (size_t)Q * k : 0;
    const size_t n_dense = mode == 1 ? (size_t)Q * ix->n_docs : 0;
    {   // plan slots: the batch's longest query (term_ptr is on the host here), within the per-call budget
        int max_nt = 1;
        for (int q = 0; q < Q; ++q) max_nt = std::max(max_nt, term_ptr[q + 1] - term_ptr[q]);
        ix->plan_t = std::min(bm25_pick_plan_t(h, ix, Q), std::min(BM_PLAN_T, max_nt));
    }
    size_t total = stage_size(Q + 1, 4) + stage_size(std::max(1, n_terms_q), 4) + stage_size(n_part, 8) + stage_size(n_part, 4) +
                   stage_size((size_t)Q * nr, 4) + 2 * stage_size(n_out, 8) + 2 * stage_size(n_out, 4) + 2 * stage_size(Q, 8) +
                   stage_size(n_out, 8) + stage_size(n_dense, 8) + stage_size(bm25_plan_off_entries(ix, Q), 4) +
                   stage_size((size_t)Q * BM_PLAN_T, sizeof(bm_plan_meta));
    if ((rc = stage_reserve(h, total))) return rc;
    char* p = (char*)h->stage;
    int32_t* tp = stage_take<int32_t>(p, Q + 1);
    int32_t* tm = stage_take<int32_t>(p, std::max(1, n_terms_q));
    bm25_topk_ws w;
    w.part_key = stage_take<uint64_t>(p, n_part);
    w.part_row = stage_take<uint32_t>(p, n_part);
    w.part_cnt = stage_take<int>(p, (size_t)Q * nr);
    w.run_key = stage_take<uint64_t>(p, n_out);
    w.run_row = stage_take<uint32_t>(p, n_out);
    w.tau = stage_take<uint64_t>(p, Q);
    int64_t* idd = stage_take<int64_t>(p, n_out);
    int32_t* rwd = stage_take<int32_t>(p, n_out);
    double* mxd = stage_take<double>(p, Q);
    double* scd = stage_take<double>(p, n_out);
    double* dd = stage_take<double>(p, n_dense);
    w.plan.off = stage_take<int32_t>(p, bm25_plan_off_entries(ix, Q));
    w.plan.meta = stage_take<bm_plan_meta>(p, (size_t)Q * BM_PLAN_T);       // (sized for the largest plan_t)
    HIP_TRY(h, hipMemcpyAsync(tp, term_ptr, (size_t)(Q + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (n_terms_q) HIP_TRY(h, hipMemcpyAsync(tm, terms, (size_t)n_terms_q * sizeof(int32_t), hipMemcpyHostToDevice, st));
    if (mode == 0) {
        const bool aligned = ix == h->bm25 && h->n_rows == ix->n_docs;
        const bm25_topk_out o = {aligned ? h->ids : (const int64_t*)nullptr, aligned ? h->id_base : (int64_t)0, idd, rwd, scd, mxd,
                                 ix->normalize};
        bm25_launch_topk(h, ix, tp, tm, Q, k, w, o, tenants, tenant, st);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(ids_out, idd, n_out * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        if (rows_out) HIP_TRY(h, hipMemcpyAsync(rows_out, rwd, n_out * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(h, hipMemcpyAsync(scores_out, scd, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
        if (raw_max_out) HIP_TRY(h, hipMemcpyAsync(raw_max_out, mxd, (size_t)Q * sizeof(double), hipMemcpyDeviceToHost, st));
    } else {
        bm25_launch_plan(ix, tp, tm, Q, w.plan, st);
        BM_RANGE_LAUNCH(h, ix, nr, Q, st, (bm_dense_extra{nullptr, 0, nullptr}), nr, tp, tm, k, 1, dd, (uint64_t*)nullptr, (uint32_t*)nullptr, 0,
                        (const uint64_t*)nullptr, (int*)nullptr, (const int32_t*)nullptr, -1, (const int32_t*)w.plan.off,
                        (const bm_plan_meta*)w.plan.meta)
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipMemcpyAsync(dense_out, dd, n_dense * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(h, hipStreamSynchronize(st));
    return RAG_OK;
}

static int bm25_ensure_plan(rag_ctx* h, rag_bm25_index* ix, int Q) {
    ix->plan_t = bm25_pick_plan_t(h, ix, Q);
    const size_t need = bm25_plan_off_entries(ix, Q);
    if (need > ix->ws_plan_entries) {
        hipFree(ix->ws_plan_off);
        ix->ws_plan_off = nullptr;
        ix->ws_plan_entries = 0;
        HIP_TRY(h, hipMalloc(&ix->ws_plan_off, need * sizeof(int32_t)));
        ix->ws_plan_entries = need;
    }
    if (Q > ix->ws_plan_q) {
        hipFree(ix->ws_plan_meta);
        ix->ws_plan_meta = nullptr;
        ix->ws_plan_q = 0;
        HIP_TRY(h, hipMalloc(&ix->ws_plan_meta, (size_t)Q * BM_PLAN_T * sizeof(bm_plan_meta)));
        ix->ws_plan_q = Q;
    }
    return RAG_OK;
}

// device-pointer entry: everything stays in HBM, asynchronous on `st` (workspace grows on first use / larger Q)
static int bm25_topk_dev_batch(rag_ctx* h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k, int tenant, int64_t* ids_dev,
                               int32_t* rows_dev, double* scores_dev, double* raw_max_dev, hipStream_t st);

// The per-call workspace (partial lists [Q][n_ranges][k] x 12 B + the plan) grows with Q x shard size: on a 12.5M-document shard a
// query takes ~9 MB at k = 100. Batches are therefore run in sub-batches whose workspace stays under BM_WS_BUDGET (queries are
// independent: same results; the 1M-document bench index runs 1,024 queries in one piece, the 12.5M-document shard 256).
#define BM_WS_BUDGET ((size_t)6 << 30)
int bm25_topk_dev(rag_ctx* h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k, int tenant, int64_t* ids_dev,
                  int32_t* rows_dev, double* scores_dev, double* raw_max_dev, hipStream_t st) {
    ARG_CHECK(h, h->bm25 != nullptr, "no BM25 index loaded");
    ARG_CHECK(h, Q > 0 && Q <= 65535 && k > 0, "bm25_topk_dev: bad arguments");
    const size_t per_query = (size_t)h->bm25->n_ranges * ((size_t)k * 12 + 4 + 8 * 4) + (size_t)k * 12;
    const size_t budget = h->opt.bm25_ws_mb > 0 ? (size_t)h->opt.bm25_ws_mb << 20 : BM_WS_BUDGET;
    const int qb = (int)std::max<size_t>(1, std::min<size_t>((size_t)Q, budget / std::max<size_t>(1, per_query)));
    for (int q0 = 0; q0 < Q; q0 += qb) {
        const int nq = std::min(qb, Q - q0);
        const int rc = bm25_topk_dev_batch(h, term_ptr_dev + q0, terms_dev, nq, k, tenant, ids_dev + (size_t)q0 * k,
                                           rows_dev ? rows_dev + (size_t)q0 * k : nullptr, scores_dev + (size_t)q0 * k,
                                           raw_max_dev ? raw_max_dev + q0 : nullptr, st);
        if (rc) return rc;
    }
    return RAG_OK;
}

static int bm25_topk_dev_batch(rag_ctx* h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, int k, int tenant, int64_t* ids_dev,
                               int32_t* rows_dev, double* scores_dev, double* raw_max_dev, hipStream_t st) {
    ARG_CHECK(h, h->bm25 != nullptr, "no BM25 index loaded");
    ARG_CHECK(h, Q > 0 && Q <= 65535 && term_ptr_dev && ids_dev && scores_dev, "bm25_topk_dev: bad arguments");
    ARG_CHECK(h, k > 0 && k <= BM_MERGE / 2 && k <= BM_RANGE, "bm25: 0 < k <= 1024");
    rag_bm25_index* ix = h->bm25;
    const int32_t* tenants = nullptr;
    int rc = bm25_tenant_args(h, ix, tenant, &tenants);
    if (rc) return rc;
    if ((rc = bm25_set_attr(h))) return rc;
    const size_t need = (size_t)Q * ix->n_ranges * k;
    if (need > ix->ws_entries) {
        hipFree(ix->ws_key); hipFree(ix->ws_row);
        ix->ws_key = nullptr; ix->ws_row = nullptr; ix->ws_entries = 0;
        HIP_TRY(h, hipMalloc(&ix->ws_key, need * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&ix->ws_row, need * sizeof(uint32_t)));
        ix->ws_entries = need;
    }
    if (Q > ix->ws_tau_q) {
        hipFree(ix->ws_tau);
        ix->ws_tau = nullptr;
        ix->ws_tau_q = 0;
        HIP_TRY(h, hipMalloc(&ix->ws_tau, (size_t)Q * sizeof(uint64_t)));
        ix->ws_tau_q = Q;
    }
    const size_t need_cnt = (size_t)Q * ix->n_ranges;
    if (need_cnt > ix->ws_cnt_n) {
        hipFree(ix->ws_cnt);
        ix->ws_cnt = nullptr;
        ix->ws_cnt_n = 0;
        HIP_TRY(h, hipMalloc(&ix->ws_cnt, need_cnt * sizeof(int)));
        ix->ws_cnt_n = need_cnt;
    }
    const size_t need_run = (size_t)Q * k;
    if (need_run > ix->ws_run_n) {
        hipFree(ix->ws_run_key); hipFree(ix->ws_run_row);
        ix->ws_run_key = nullptr; ix->ws_run_row = nullptr; ix->ws_run_n = 0;
        HIP_TRY(h, hipMalloc(&ix->ws_run_key, need_run * sizeof(uint64_t)));
        HIP_TRY(h, hipMalloc(&ix->ws_run_row, need_run * sizeof(uint32_t)));
        ix->ws_run_n = need_run;
    }
    if ((rc = bm25_ensure_plan(h, ix, Q))) return rc;
    // doc ids follow the dense index's mapping when both indexes cover the same rows (hybrid fusion needs one id space)
    const bool aligned = h->n_rows == ix->n_docs;
    const bm25_topk_ws w = {ix->ws_key, ix->ws_row, ix->ws_cnt, ix->ws_run_key, ix->ws_run_row, ix->ws_tau, {ix->ws_plan_off, ix->ws_plan_meta}};
    const bm25_topk_out o = {aligned ? h->ids : (const int64_t*)nullptr, aligned ? h->id_base : (int64_t)0, ids_dev, rows_dev, scores_dev,
                             raw_max_dev, ix->normalize};
    if ((rc = prof_begin(h, 1, st))) return rc;
    bm25_launch_topk(h, ix, term_ptr_dev, terms_dev, Q, k, w, o, tenants, tenant, st);
    HIP_TRY(h, hipGetLastError());
    return prof_end(h, 1, st);
}

// raw float64 scores of EVERY document for Q queries, device pointers, asynchronous: out_dev[Q][n_docs]
// (the all-document BM25Okapi.get_scores of rag/retrieval.py:340-341, for the index-level linear fusion)
// raw32_dev / ld / max_key_dev (all optional): see bm_dense_extra; `tenant` restricts the maximum to the tenant's documents
int bm25_scores_dev(rag_ctx* h, const int32_t* term_ptr_dev, const int32_t* terms_dev, int Q, double* out_dev, hipStream_t st,
                    float* raw32_dev, int64_t ld, unsigned long long* max_key_dev, int tenant) {
    ARG_CHECK(h, h->bm25 != nullptr, "no BM25 index loaded");
    ARG_CHECK(h, Q > 0 && Q <= 65535 && term_ptr_dev && out_dev, "bm25_scores_dev: bad arguments");
    rag_bm25_index* ix = h->bm25;
    int rc = bm25_set_attr(h);
    if (rc) return rc;
    const int32_t* tenants = nullptr;
    if (max_key_dev != nullptr && (rc = bm25_tenant_args(h, ix, tenant, &tenants))) return rc;
    const bm_dense_extra dx = {raw32_dev, ld, max_key_dev};
    if ((rc = bm25_ensure_plan(h, ix, Q))) return rc;
    const bm25_plan_ws pw = {ix->ws_plan_off, ix->ws_plan_meta};
    bm25_launch_plan(ix, term_ptr_dev, terms_dev, Q, pw, st);
    BM_RANGE_LAUNCH(h, ix, ix->n_ranges, Q, st, dx, ix->n_ranges, term_ptr_dev, terms_dev, 1, 1, out_dev, (uint64_t*)nullptr,
                    (uint32_t*)nullptr, 0, (const uint64_t*)nullptr, (int*)nullptr, tenants, tenants ? tenant : -1,
                    (const int32_t*)pw.off, (const bm_plan_meta*)pw.meta)
    HIP_TRY(h, hipGetLastError());
    return RAG_OK;
}

int bm25_set_normalize(rag_ctx* h, int on) {
    ARG_CHECK(h, h->bm25 != nullptr, "no BM25 index loaded");
    h->bm25->normalize = on ? 1 : 0;
    return RAG_OK;
}

int64_t bm25_n_docs(const rag_ctx* h) { return h->bm25 ? h->bm25->n_docs : -1; }

// HBM bytes of an index built from these postings, from the host CSR offsets alone (no device call): postings (doc + impact,
// 12 B each), per-term metadata (32 B each) and the bracket tables (4 B per entry, <= nnz bytes by construction).
// launch geometry of one scoring launch (bm_make_grid), for tests and capacity planning: out = {workgroups, ranges, queries, query groups
// per range, queries per group (0: range-major order)}
int bm25_grid_plan(int n_ranges_in_launch, int n_queries, int linear, int64_t* out5) {
    if (!out5 || n_ranges_in_launch <= 0 || n_queries <= 0) return RAG_ERR_ARG;
    const bm_grid g = bm_make_grid(n_ranges_in_launch, n_queries, linear);
    out5[0] = g.blocks; out5[1] = g.nr_l; out5[2] = g.n_queries; out5[3] = g.n_groups; out5[4] = g.qgroup_len;
    return RAG_OK;
}

int bm25_index_bytes(const int64_t* indptr, int64_t n_docs, int64_t n_terms, int64_t* postings_out, int64_t* meta_out, int64_t* table_out) {
    if (!indptr || n_docs <= 0 || n_terms < 0) return RAG_ERR_ARG;
    const int64_t n_pad = (n_docs + BM_RANGE - 1) / BM_RANGE * BM_RANGE;
    int64_t n_tab = 0;
    for (int64_t t = 0; t < n_terms; ++t) {
        const int64_t df = indptr[t + 1] - indptr[t];
        if (df < 0) return RAG_ERR_ARG;
        int64_t e_t = 0;
        bm_plan_term(df, n_pad, &e_t);
        n_tab += e_t;
    }
    if (postings_out) *postings_out = ((n_terms ? indptr[n_terms] - indptr[0] : 0) + 8) * 12;     // default form: doc i32 + impact f64
    if (meta_out) *meta_out = n_terms * (int64_t)sizeof(bm_term_meta);
    if (table_out) *table_out = n_tab * 4;
    return RAG_OK;
}

int bm25_topk_host(rag_ctx* h, const int32_t* term_ptr, const int32_t* terms, int Q, int k, int tenant, int64_t* ids_out,
                   int32_t* rows_out, double* scores_out, double* raw_max_out) {
    ARG_CHECK(h, ids_out && scores_out, "bm25_topk: null output");
    return bm25_run(h, h->bm25, term_ptr, terms, Q, k, 0, tenant, ids_out, rows_out, scores_out, raw_max_out, nullptr);
}

int bm25_scores_host(rag_ctx* h, const int32_t* term_ptr, const int32_t* terms, int Q, double* out) {
    ARG_CHECK(h, out, "bm25_scores: null output");
    return bm25_run(h, h->bm25, term_ptr, terms, Q, 1, 1, -1, nullptr, nullptr, nullptr, nullptr, out);
}

// Stateless scoring of an AD-HOC corpus (HybridRetriever.hybrid_search builds a fresh BM25Okapi over the corpus it is handed on
// every call, rag/retrieval.py:333-341): the postings are uploaded, scored and dropped inside the call; the resident
// postings of the index (rag_bm25_load_host) are not touched.
int bm25_scores_adhoc_host(rag_ctx* h, const int64_t* indptr, const int32_t* doc, const int32_t* tf, const int32_t* doc_len,
                           const double* idf, int64_t n_docs, int64_t n_terms, double avgdl, double k1, double b,
                           const int32_t* term_ptr, const int32_t* terms, int Q, double* out) {
    ARG_CHECK(h, out, "bm25_scores_adhoc: null output");
    rag_bm25_index* ix = nullptr;
    int rc = bm25_build(h, indptr, doc, tf, doc_len, idf, n_docs, n_terms, avgdl, k1, b, &ix);
    if (rc) return rc;
    rc = bm25_run(h, ix, term_ptr, terms, Q, 1, 1, -1, nullptr, nullptr, nullptr, nullptr, out);
    bm25_index_free(ix);
    return rc;
}
