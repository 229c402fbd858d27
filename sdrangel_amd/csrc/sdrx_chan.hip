// libsdrx.so: sdrx_chan_* -- a bank of DownChannelizers fed from one device stream
// (reference: sdrbase/dsp/downchannelizer.{h,cpp}).  Host planner + launches; kernel in tree_kernel.hpp.
#include "sdrx_common.hpp"
#include "tree_kernel.hpp"
#include <vector>
#include <array>
#include <cstring>
#include <cstdlib>
#include <new>
#include <algorithm>

using namespace sdrx;

/* ------------------------------------------------------------------ the float bisection
 * DownChannelizer::applyConfiguration / createFilterChain (downchannelizer.cpp:157-189, 250-287),
 * restated.  All interval arithmetic is float32 (`Real`); the reference writes `x / 2.0` in two
 * places, which is evaluated in double and rounded to float when passed on -- kept.               */
static bool contains(float ss, float se, float cs, float ce)
{
    if (se <= ss || ce <= cs) return false;               // signalContainsChannel (:240-248)
    return ss <= cs && se >= ce;
}

static int plan_chain(int32_t in_rate, int32_t req_rate, int32_t req_fc, uint8_t* modes, int cap,
                      int32_t* out_rate, int32_t* ofs_out)
{
    if (in_rate == 0) { *out_rate = 0; *ofs_out = 0; return 0; }   // "m_inputSampleRate=0 aborting"
    float s = (float)(in_rate / -2), e = (float)(in_rate / 2);
    const float cs = (float)(req_fc - req_rate / 2), ce = (float)(req_fc + req_rate / 2);
    int n = 0;
    while (n < cap) {
        const float bw = e - s, rot = bw / 4;
        const float mid_lo = (float)((double)s + (double)bw / 2.0);       // sigStart + sigBw / 2.0
        const float mid_hi = e - bw / 2.0f;                               // sigEnd - sigBw / 2.0f
        if (contains(s, mid_lo, cs, ce)) { modes[n++] = SDRX_MODE_LOWER; e = mid_lo; continue; }
        if (contains(mid_hi, e, cs, ce)) { modes[n++] = SDRX_MODE_UPPER; s = mid_hi; continue; }
        const float cs2 = s + rot, ce2 = e - rot;
        if (contains(cs2, ce2, cs, ce)) { modes[n++] = SDRX_MODE_CENTER; s = cs2; e = ce2; continue; }
        break;
    }
    const float ofs = (float)(((double)(ce - cs) / 2.0 + (double)cs) - ((double)(e - s) / 2.0 + (double)s));
    *ofs_out = (int32_t)ofs;                               // Real -> int m_currentCenterFrequency
    *out_rate = in_rate / (1 << n);
    return n;
}

/* ------------------------------------------------------------------ host-side plan of one group */
namespace {

constexpr int MAX_STAGES = 30;
constexpr int LDS_BUDGET_DW_DEFAULT = 40 * 1024 / 4;       // four workgroups per CU (the kernel has no static LDS)
// Wide banks (cfg 4: 256 channels) have a dense tree top whose levels need ~25 KB each; with 40 KB the greedy cut ends up
// with 1-2 levels per pass and six passes.  64 KB (two workgroups per CU) measured 1.29 vs 1.42 ms per 64 Mi-sample feed
// for 256 channels, but 0.82 vs 0.71 ms for 128 and 0.62 vs 0.45 ms for 32 -- so only wide banks get it.
static int lds_budget_dw(size_t n_channels)
{
    const char* e = getenv("SDRX_CHAN_LDS_KB");
    if (e && atoi(e) >= 16 && atoi(e) <= 150) return atoi(e) * 1024 / 4;
    return n_channels >= 192 ? 64 * 1024 / 4 : LDS_BUDGET_DW_DEFAULT;
}
constexpr int LDS_HARD_DW = 150 * 1024 / 4;
// levels per pass: 6 = one warm-up chunk per segment.  Deeper passes (experiment, DESIGN 4.3: fewer node-stream bytes
// for more LDS) need ceil(46 * (2^levels - 1) / 4096) warm-up chunks and as many more chunks of stream history.
static int max_levels()
{
    const char* e = getenv("SDRX_CHAN_MAX_LEVELS");
    if (e && atoi(e) >= 1 && atoi(e) <= TK_MAX_LEVELS) return atoi(e);
    return TK_DEFAULT_LEVELS;
}             // < the 159 KB of dynamic LDS the kernel may ask for

// Half-band engine of the bank's kernels: the matrix cores (i8 MFMA, hb_mfma.hpp) unless SDRX_CHAN_ENGINE=valu asks for the
// dot2 kernel of rounds 1-2 (both are gfx950 code, both bit-exact; tests run the matrix under each).  Read at plan time.
static bool engine_mfma()
{
    const char* e = getenv("SDRX_CHAN_ENGINE");
    return !(e && strcmp(e, "valu") == 0);
}
// a level runs on the matrix cores when an entry fills whole tiles (16 blocks of 16 outputs per component and chunk)
static bool level_is_mfma(bool engine, int rel) { return engine && (TK_CHUNK >> rel) >= 256; }

struct HNode {
    int parent = -1, mode = 0, depth = 0;
    int child[3] = { -1, -1, -1 };
    std::vector<int> ends;        // bank channel ids whose chain ends here
    int stream = -1;              // index into Group::streams if this node's output is a global stream
};

struct Channel {
    int32_t req_rate = 0, req_fc = 0, out_rate = 0, ofs = 0;
    int n = 0;
    uint8_t modes[32] = { 0 };
    int group = -1;               // -1: pass-through (0 stages) or dead
    bool passthrough = false;
    bool dead = false;            // removed: the index stays reserved, nothing is produced any more
    DevBuf out;                   // device queue of packed Samples
    int64_t avail = 0;            // complex samples queued
    int64_t last_off = 0, last_n = 0;
    int sink = -1;                // sink index inside its group
};

struct Stream {
    int trie_node = 0, depth = 0, pass = 0, subtree = -1;
    uint32_t* hist[2] = { nullptr, nullptr };
    long hist_len = TK_HIST;      // samples kept between feeds: (warm-up chunks of its subtree + 1) chunks
    int cur = 0;
    DevBuf mid;                   // new samples of a node stream (unused for the raw stream)
    int sink = -1;                // sink (in the producing pass) that fills `mid`
};

struct SinkInfo { int kind; int ch; int stream; int depth; int next; };   // kind 0 channel, 1 node stream

struct Group {
    int index = -1;               // position in sdrx_chan_bank::groups
    int64_t T = 0;                // samples fed since this group's epoch
    std::vector<int> chans;
    std::vector<HNode> trie;
    std::vector<Stream> streams;  // [0] = the raw stream
    std::vector<std::vector<int>> passes;
    std::vector<TkSubtree> subtrees;
    std::vector<TkNode> nodes;
    std::vector<TkArray> arrays;
    std::vector<TkMJob> mjobs;    // matrix-core jobs of every MFMA level (tree_kernel.hpp)
    std::vector<SinkInfo> sinks;
    int max_lds_dw = 0;
    bool mfma = true;             // engine this group was planned for
    void* d_static = nullptr;     // subtrees | nodes | arrays
    TkSubtree* d_subtrees = nullptr; TkNode* d_nodes = nullptr; TkArray* d_arrays = nullptr; TkMJob* d_mjobs = nullptr;
};

int arm_len(int rel_depth) { return HIST / 2 + (TK_CHUNK >> (rel_depth + 2)); }   // dwords

// LDS dwords a subtree of `levels` levels below trie node `root` needs: two arm regions (even / odd producer
// level, each as large as its biggest level), 16 dwords of persistent history per array, the node table
// (a lower/upper sibling pair shares one entry)
int subtree_lds(const std::vector<HNode>& trie, int root, int levels, int* n_nodes_out)
{
    int region[2] = { 0, 0 }, n_arrays = 0, n_entries = 0, n_sinks = 0;
    std::vector<int> cur{ root };
    for (int rel = 0; rel < levels; rel++) {
        std::vector<int> nxt;
        int level_dw = 0;
        for (int id : cur) {
            bool c = trie[id].child[0] >= 0, lu = trie[id].child[1] >= 0 || trie[id].child[2] >= 0;
            if (!c && !lu) continue;
            const int na = 2 + (c ? 2 : 0) + (lu ? 2 : 0);
            level_dw += arm_len(rel) * na; n_arrays += na;
            n_entries += (c ? 1 : 0) + (lu ? 1 : 0);
            for (int m = 0; m < 3; m++) if (trie[id].child[m] >= 0) {
                const int kid = trie[id].child[m];
                nxt.push_back(kid);
                // sinks of the stage: every channel that ends there, plus a node stream where the tree goes on below the pass
                n_sinks += (int)trie[kid].ends.size();
                if (rel + 1 == levels && (trie[kid].child[0] >= 0 || trie[kid].child[1] >= 0 || trie[kid].child[2] >= 0)) n_sinks++;
            }
        }
        region[rel & 1] = std::max(region[rel & 1], level_dw);
        cur.swap(nxt);
        if (cur.empty()) break;
    }
    *n_nodes_out = n_entries;
    // + one table dword per array (+ 1 for the 8-byte alignment of what follows) + the sink descriptors
    return region[0] + region[1] + n_arrays * 16 + n_entries * TK_NODE_DW + n_arrays + 1 + n_sinks * TK_SINK_DW;
}

int height(const std::vector<HNode>& trie, int id)
{
    int h = 0;
    for (int m = 0; m < 3; m++) if (trie[id].child[m] >= 0) h = std::max(h, 1 + height(trie, trie[id].child[m]));
    return h;
}

} // namespace

struct sdrx_chan_bank {
    int device = 0, cus = 256;
    int32_t in_rate = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    std::vector<Channel> ch;
    std::vector<Group*> groups;
    DevBuf stage_in;              // host-pointer feeds are staged here
    DevBuf scratch;               // queue compaction
    // per-feed dynamic tables: pinned host ring + device copies
    static constexpr int RING = 4;
    void* h_dyn[RING] = { nullptr, nullptr, nullptr, nullptr };
    void* d_dyn[RING] = { nullptr, nullptr, nullptr, nullptr };
    size_t dyn_cap[RING] = { 0, 0, 0, 0 };
    hipEvent_t dyn_ev[RING] = { nullptr, nullptr, nullptr, nullptr };
    int dyn_next = 0;
    char last_name[96] = "";
    int last_grid = 0, last_block = 0, last_lds = 0;
    EventTimer timer;
};

static void free_group(Group* g)
{
    if (!g) return;
    for (auto& s : g->streams) { for (int i = 0; i < 2; i++) if (s.hist[i]) (void)hipFree(s.hist[i]); s.mid.release(); }
    if (g->d_static) (void)hipFree(g->d_static);
    delete g;
}

// Build trie, cut into passes, lay out LDS, fill the static device tables.
static int plan_group(sdrx_chan_bank* b, Group* g)
{
    g->trie.clear(); g->trie.emplace_back();
    for (int c : g->chans) {
        Channel& ch = b->ch[c];
        int id = 0;
        for (int s = 0; s < ch.n; s++) {
            const int m = ch.modes[s];
            if (g->trie[id].child[m] < 0) {
                HNode nn; nn.parent = id; nn.mode = m; nn.depth = s + 1;
                g->trie.push_back(nn);
                g->trie[id].child[m] = (int)g->trie.size() - 1;
            }
            id = g->trie[id].child[m];
        }
        g->trie[id].ends.push_back(c);
    }

    g->streams.clear(); g->passes.clear(); g->subtrees.clear(); g->nodes.clear(); g->arrays.clear(); g->sinks.clear(); g->mjobs.clear();
    g->mfma = engine_mfma();
    { Stream raw; raw.trie_node = 0; raw.depth = 0; raw.pass = 0; g->streams.push_back(std::move(raw)); g->trie[0].stream = 0; }

    for (size_t si = 0; si < g->streams.size(); si++) {
        const int root = g->streams[si].trie_node, pass = g->streams[si].pass;
        const int h = height(g->trie, root);
        if (h == 0) { g->streams[si].subtree = -1; continue; }
        int levels = 1, nn = 0;
        while (levels < std::min(h, max_levels()) && subtree_lds(g->trie, root, levels + 1, &nn) <= lds_budget_dw(g->chans.size())) levels++;
        int lds_need = subtree_lds(g->trie, root, levels, &nn);
        if (lds_need > LDS_HARD_DW) { set_error("channel tree does not fit LDS"); return SDRX_EINVAL; }

        TkSubtree st; memset(&st, 0, sizeof st);
        st.n_levels = levels;
        st.sink_base = (int)g->sinks.size();
        st.warm = (int)((46L * ((1L << levels) - 1) + TK_CHUNK - 1) / TK_CHUNK);
        if (st.warm < 1) st.warm = 1;
        g->streams[si].hist_len = (long)(st.warm + 1) * TK_CHUNK;
        st.node_base = (int)g->nodes.size();
        st.array_base = (int)g->arrays.size();
        // window offsets are assigned per producer level inside region (level & 1); fixed up to absolute LDS offsets below
        int reg_used[2] = { 0, 0 }, reg_size[2] = { 0, 0 }, cur_level = 0;
        struct Arms { int E[2], O[2], A[2]; };
        auto alloc_arms = [&](int id, int rel, bool inner) {
            Arms a; a.E[0] = a.E[1] = a.O[0] = a.O[1] = a.A[0] = a.A[1] = -1;
            const bool c = g->trie[id].child[0] >= 0, lu = g->trie[id].child[1] >= 0 || g->trie[id].child[2] >= 0;
            if (!inner || (!c && !lu)) return a;
            const int len = arm_len(rel);
            if (rel != cur_level) { cur_level = rel; reg_used[rel & 1] = 0; }
            // returns the INDEX of the array (relative to the subtree's list); node fields are patched to offsets later
            auto take = [&](bool odd_arm = false) {
                const int idx = (int)g->arrays.size() - st.array_base;
                // an odd arm read by an MFMA level holds x ^ 0x0080 (hb_mfma.hpp): the consumers of an array sit one level down
                const int bias = odd_arm && level_is_mfma(g->mfma, rel + 1) ? 1 : 0;
                g->arrays.push_back(TkArray{ reg_used[rel & 1], len, rel & 1, bias });   // store: region id for now
                reg_used[rel & 1] += len; reg_size[rel & 1] = std::max(reg_size[rel & 1], reg_used[rel & 1]);
                return idx;
            };
            a.E[0] = take(); a.E[1] = take();
            if (c) { a.O[0] = take(true); a.O[1] = take(true); }
            if (lu) { a.A[0] = take(true); a.A[1] = take(true); }
            return a;
        };
        std::vector<int> cur{ root };
        std::vector<Arms> cur_arms{ alloc_arms(root, 0, true) };
        const Arms root_arms = cur_arms[0];
        st.root_arr_cnt = (int)g->arrays.size() - st.array_base;
        int rel_nodes = 0;
        for (int rel = 1; rel <= levels; rel++) {
            std::vector<int> nxt; std::vector<Arms> nxt_arms;
            TkLevel& lv = st.lv[rel - 1];
            lv.node_base = rel_nodes;
            lv.nout = TK_CHUNK >> rel;
            lv.r_log2 = 3; lv.jobs_log2 = 0;                           // fixed up below once the level's entry count is known
            lv.mfma = level_is_mfma(g->mfma, rel) ? 1 : 0;
            lv.mjob_base = 0; lv.n_mjobs = 0;
            lv.xm = rel < levels && level_is_mfma(g->mfma, rel + 1) ? HBM_BIAS2 : 0u;
            lv.arr_base = (int)g->arrays.size() - st.array_base;
            int n_entries = 0;
            for (size_t pi = 0; pi < cur.size(); pi++) {
                const Arms pa = cur_arms[pi];
                int kid[3]; Arms own[3];
                for (int m = 0; m < 3; m++) {
                    kid[m] = g->trie[cur[pi]].child[m];
                    if (kid[m] < 0) continue;
                    own[m] = alloc_arms(kid[m], rel, rel < levels);
                    nxt.push_back(kid[m]); nxt_arms.push_back(own[m]);
                }
                // one stage's output side: arms, centre taps, sinks (channel ends; a node stream if the tree goes on below this pass)
                auto fill = [&](TkOut& o, int m) {
                    const int id = kid[m];
                    o.present = 1;
                    o.outE_I = own[m].E[0]; o.outE_Q = own[m].E[1];
                    o.outO_I = own[m].O[0]; o.outO_Q = own[m].O[1];
                    o.outA_I = own[m].A[0]; o.outA_Q = own[m].A[1];
                    if (m == SDRX_MODE_CENTER) { o.cIe = pk16(0, 2048); o.cIo = pk16(2048, 0); o.cQe = pk16(0, 2048); o.cQo = pk16(2048, 0); }
                    else {
                        const int sg = m == SDRX_MODE_LOWER ? 1 : -1;
                        // lower: k odd -> (-im, re), k even -> (im, -re); upper: the negation
                        o.cIo = pk16(-2048 * sg, 0); o.cQo = pk16(2048 * sg, 0);
                        o.cIe = pk16(0, 2048 * sg);  o.cQe = pk16(0, -2048 * sg);
                    }
                    o.sink = -1;
                    for (int c : g->trie[id].ends) {
                        SinkInfo sk{ 0, c, -1, g->trie[id].depth, o.sink };
                        g->sinks.push_back(sk); o.sink = (int)g->sinks.size() - 1;
                        b->ch[c].sink = o.sink;
                    }
                    const bool has_kids = g->trie[id].child[0] >= 0 || g->trie[id].child[1] >= 0 || g->trie[id].child[2] >= 0;
                    if (rel == levels && has_kids) {
                        Stream ms; ms.trie_node = id; ms.depth = g->trie[id].depth; ms.pass = pass + 1;
                        SinkInfo sk{ 1, -1, (int)g->streams.size(), g->trie[id].depth, o.sink };
                        g->sinks.push_back(sk); o.sink = (int)g->sinks.size() - 1;
                        ms.sink = o.sink;
                        g->trie[id].stream = (int)g->streams.size();
                        g->streams.push_back(std::move(ms));
                    }
                };
                if (kid[SDRX_MODE_CENTER] >= 0) {
                    TkNode nd; memset(&nd, 0xff, sizeof nd);
                    nd.oddI = pa.O[0]; nd.oddQ = pa.O[1]; nd.cenI = pa.E[0]; nd.cenQ = pa.E[1];
                    fill(nd.a, SDRX_MODE_CENTER);
                    nd.b.present = 0; nd.mode_a = SDRX_MODE_CENTER;
                    g->nodes.push_back(nd); n_entries++;
                }
                if (kid[SDRX_MODE_LOWER] >= 0 || kid[SDRX_MODE_UPPER] >= 0) {
                    // lower and upper siblings read the same alternating-sign odd arm and differ only in the centre tap:
                    // fused into one entry (a = first present, b = the other)
                    TkNode nd; memset(&nd, 0xff, sizeof nd);
                    nd.oddI = pa.A[0]; nd.oddQ = pa.A[1]; nd.cenI = pa.E[1]; nd.cenQ = pa.E[0];   // I <- eQ, Q <- eI
                    nd.b.present = 0;
                    if (kid[SDRX_MODE_LOWER] >= 0) {
                        fill(nd.a, SDRX_MODE_LOWER); nd.mode_a = SDRX_MODE_LOWER;
                        if (kid[SDRX_MODE_UPPER] >= 0) fill(nd.b, SDRX_MODE_UPPER);
                    } else { fill(nd.a, SDRX_MODE_UPPER); nd.mode_a = SDRX_MODE_UPPER; }
                    g->nodes.push_back(nd); n_entries++;
                }
            }
            rel_nodes += n_entries;
            lv.arr_cnt = (int)g->arrays.size() - st.array_base - lv.arr_base;
            lv.n_nodes = n_entries;
            {   // outputs per job: 8 while that gives every lane of the workgroup a job, else 4, else 2 (a level costs one job time)
                int rl = 3;
                const char* er = getenv("SDRX_CHAN_R8");
                if (!er) while (rl > 1 && (long)n_entries * (lv.nout >> rl) < TK_THREADS) rl--;
                lv.r_log2 = rl;
                int jl = 0; while (((1 << rl) << jl) < lv.nout) jl++;
                lv.jobs_log2 = jl;
            }
            cur.swap(nxt); cur_arms.swap(nxt_arms);
        }
        st.n_nodes = rel_nodes;
        st.n_arrays = (int)g->arrays.size() - st.array_base;
        // absolute layout: [region 0][region 1][history store: 16 dwords per array][node table]
        const int reg_base[2] = { 0, reg_size[0] };
        const int store_base = reg_size[0] + reg_size[1];
        for (int i = 0; i < st.n_arrays; i++) {
            TkArray& a = g->arrays[(size_t)(st.array_base + i)];
            a.off += reg_base[a.store]; a.store = store_base + 16 * i;
        }
        auto fix = [&](int& v) { if (v >= 0) v = g->arrays[(size_t)(st.array_base + v)].off; };
        st.root_xm = level_is_mfma(g->mfma, 1) ? HBM_BIAS2 : 0u;
        { const char* e = getenv("SDRX_CHAN_DBG"); st.dbg = e ? atoi(e) : 0; }
        { Arms r = root_arms; for (int q = 0; q < 2; q++) { fix(r.E[q]); fix(r.O[q]); fix(r.A[q]); }
          st.rootE_I = r.E[0]; st.rootE_Q = r.E[1]; st.rootO_I = r.O[0]; st.rootO_Q = r.O[1]; st.rootA_I = r.A[0]; st.rootA_Q = r.A[1]; }
        for (int i = 0; i < rel_nodes; i++) {
            TkNode& nd = g->nodes[(size_t)(st.node_base + i)];
            fix(nd.oddI); fix(nd.oddQ); fix(nd.cenI); fix(nd.cenQ);
            for (TkOut* o : { &nd.a, &nd.b }) {
                if (!o->present) continue;
                fix(o->outE_I); fix(o->outE_Q); fix(o->outO_I); fix(o->outO_Q); fix(o->outA_I); fix(o->outA_Q);
            }
        }
        for (int i = 0; i < st.n_arrays; i++) {
            // the matrix-core levels read their windows as aligned 16-byte vectors
            const TkArray& a = g->arrays[(size_t)(st.array_base + i)];
            if (a.bias && (a.off & 3)) { set_error("internal: MFMA window not 16-byte aligned"); return SDRX_EINVAL; }
        }
        st.store_base = store_base;
        st.node_tab = store_base + 16 * st.n_arrays;
        st.n_sinks = (int)g->sinks.size() - st.sink_base;
        st.sink_tab = (st.node_tab + rel_nodes * TK_NODE_DW + 1) & ~1;      // 8-byte aligned: read as uint2
        st.lds_dwords = st.sink_tab + st.n_sinks * TK_SINK_DW;
        {   // matrix-core jobs: LDS byte addresses of every array a job touches, 256 tb outputs into the chunk; 64 dwords of
            // scratch take the stores to arm arrays a child does not have
            const int trash = (st.lds_dwords + 3) & ~3;
            bool any = false;
            for (int l = 0; l < levels; l++) {
                TkLevel& lv = st.lv[l];
                if (!lv.mfma) continue;
                any = true;
                lv.mjob_base = (int)g->mjobs.size();
                const int tpe = lv.nout / 256;
                for (int e = 0; e < lv.n_nodes; e++) {
                    const TkNode& nd = g->nodes[(size_t)(st.node_base + lv.node_base + e)];
                    for (int tb = 0; tb < tpe; tb++) {
                        TkMJob j; memset(&j, 0, sizeof j);
                        j.bI = (nd.oddI + 4 + 128 * tb) * 4; j.bQ = (nd.oddQ + 4 + 128 * tb) * 4;
                        j.cI = (nd.cenI + 10 + 128 * tb) * 4; j.cQ = (nd.cenQ + 10 + 128 * tb) * 4;
                        j.mode = nd.mode_a == SDRX_MODE_CENTER ? 0 : 1;
                        j.out0 = 256 * tb;
                        auto put = [&](TkMOut& m, const TkOut* o) {
                            auto at = [&](int off) { return (off >= 0 ? off + HIST / 2 + 64 * tb : trash) * 4; };
                            if (!o) { m.E_I = m.E_Q = m.O_I = m.O_Q = m.A_I = m.A_Q = trash * 4; m.sink = -1; m.flags = 0; return; }
                            m.E_I = at(o->outE_I); m.E_Q = at(o->outE_Q); m.O_I = at(o->outO_I); m.O_Q = at(o->outO_Q);
                            m.A_I = at(o->outA_I); m.A_Q = at(o->outA_Q);
                            m.sink = o->sink; m.flags = (o->outE_I >= 0 ? 1 : 0) | (o->outO_I >= 0 ? 2 : 0) | (o->outA_I >= 0 ? 4 : 0);
                        };
                        if (nd.mode_a == SDRX_MODE_UPPER) { put(j.o[0], nullptr); put(j.o[1], &nd.a); }
                        else { put(j.o[0], &nd.a); put(j.o[1], nd.b.present ? &nd.b : nullptr); }
                        // the common inner job -- a lower/upper pair whose two children are inner nodes with ONE kind of odd arm and no sink --
                        // gets a branch-free epilogue (tree_kernel.hpp): the odd target moves into O_I / O_Q whatever its kind, `kinds` says which
                        // children want the alternating-sign copy
                        auto one_odd = [](const TkMOut& m) { return m.sink < 0 && (m.flags == (1 | 2) || m.flags == (1 | 4)); };
                        // (the same treatment for single-child pairs and centre stages measured SLOWER, 3.22 vs 3.13 ms: three more inlined store groups
                        // in both the paired and the single job body)
                        // (nor did sending single-child pairs down this path with the absent child's stores going to the scratch slot: 3.15 vs 3.13)
                        const int fast = j.mode && one_odd(j.o[0]) && one_odd(j.o[1]) ? 1 : 0;
                        if (fast) {
                            j.fast = fast; j.kinds = ((j.o[0].flags & 4) ? 1 : 0) | ((j.o[1].flags & 4) ? 2 : 0);
                            for (TkMOut* m : { &j.o[0], &j.o[1] }) if (m->flags & 4) { m->O_I = m->A_I; m->O_Q = m->A_Q; }
                        }
                        g->mjobs.push_back(j);
                    }
                }
                lv.n_mjobs = (int)g->mjobs.size() - lv.mjob_base;
            }
            if (any) st.lds_dwords = trash + 64;
        }
        if (st.lds_dwords > 159 * 1024 / 4) { set_error("channel tree does not fit LDS"); return SDRX_EINVAL; }   // tables and the job scratch on top of the arm regions
        for (int l = 0; l < levels; l++) {
            st.lv[l].in_len = arm_len(l);
            // the walk's address arithmetic (tree_kernel.hpp): a level's arrays are contiguous, of one length, slots in array order
            const int pb = l == 0 ? 0 : st.lv[l - 1].arr_base, pc = l == 0 ? st.root_arr_cnt : st.lv[l - 1].arr_cnt;
            st.lv[l].prev_arr_cnt = pc;
            st.lv[l].prev_off = pc ? g->arrays[(size_t)(st.array_base + pb)].off : 0;
            st.lv[l].arr_off = st.lv[l].arr_cnt ? g->arrays[(size_t)(st.array_base + st.lv[l].arr_base)].off : 0;
            st.lv[l].arr_len = st.lv[l].arr_cnt ? g->arrays[(size_t)(st.array_base + st.lv[l].arr_base)].len : 0;
            if (pb + pc != st.lv[l].arr_base) { set_error("internal: level arrays not in order"); return SDRX_EINVAL; }
            for (int k = 0; k < st.lv[l].arr_cnt; k++) {
                const TkArray& a = g->arrays[(size_t)(st.array_base + st.lv[l].arr_base + k)];
                if (a.off != st.lv[l].arr_off + k * st.lv[l].arr_len || a.len != st.lv[l].arr_len || a.store != st.store_base + 16 * (st.lv[l].arr_base + k)) {
                    set_error("internal: level arrays not contiguous"); return SDRX_EINVAL;
                }
            }
            for (int k = 0; k < pc; k++) {
                const TkArray& a = g->arrays[(size_t)(st.array_base + pb + k)];
                if (a.off != st.lv[l].prev_off + k * st.lv[l].in_len || a.len != st.lv[l].in_len) { set_error("internal: parent arrays not contiguous"); return SDRX_EINVAL; }
            }
        }
        st.root_off = st.root_arr_cnt ? g->arrays[(size_t)st.array_base].off : 0;
        st.root_len = st.root_arr_cnt ? g->arrays[(size_t)st.array_base].len : 0;
        g->max_lds_dw = std::max(g->max_lds_dw, st.lds_dwords);
        g->streams[si].subtree = (int)g->subtrees.size();
        g->subtrees.push_back(st);
        if ((int)g->passes.size() <= pass) g->passes.resize(pass + 1);
        g->passes[pass].push_back((int)si);
        if (getenv("SDRX_CHAN_DEBUG")) {
            int nj = 0, nf = 0;
            for (int l = 0; l < levels; l++)
                for (int q = 0; q < st.lv[l].n_mjobs; q++) { nj++; nf += g->mjobs[(size_t)(st.lv[l].mjob_base + q)].fast; }
            fprintf(stderr, "sdrx plan: pass %d stream %d (trie node %d, depth %d): %d levels, %d entries, %d arrays, %d LDS dwords (regions %d + %d), %d matrix-core jobs per chunk (%d branch-free)\n",
                    pass, (int)si, root, g->streams[si].depth, levels, rel_nodes, st.n_arrays, st.lds_dwords, reg_size[0], reg_size[1], nj, nf);
        }
    }

    // device: histories + static tables
    for (auto& s : g->streams) {
        for (int i = 0; i < 2; i++) {
            SDRX_HIP(hipMalloc(reinterpret_cast<void**>(&s.hist[i]), (size_t)s.hist_len * 4));
            SDRX_HIP(hipMemsetAsync(s.hist[i], 0, (size_t)s.hist_len * 4, b->stream));
        }
    }
    const size_t b0 = g->subtrees.size() * sizeof(TkSubtree), b1 = g->nodes.size() * sizeof(TkNode), b2 = g->arrays.size() * sizeof(TkArray);
    const size_t b3 = g->mjobs.size() * sizeof(TkMJob);
    if (b0 + b1 + b2 > 0) {
        SDRX_HIP(hipMalloc(&g->d_static, b0 + b1 + b2 + b3 + 64));
        char* p = static_cast<char*>(g->d_static);
        g->d_subtrees = reinterpret_cast<TkSubtree*>(p);
        g->d_nodes = reinterpret_cast<TkNode*>(p + b0);
        g->d_arrays = reinterpret_cast<TkArray*>(p + b0 + b1);
        if (b0) SDRX_HIP(hipMemcpy(g->d_subtrees, g->subtrees.data(), b0, hipMemcpyHostToDevice));
        if (b1) SDRX_HIP(hipMemcpy(g->d_nodes, g->nodes.data(), b1, hipMemcpyHostToDevice));
        if (b2) SDRX_HIP(hipMemcpy(g->d_arrays, g->arrays.data(), b2, hipMemcpyHostToDevice));
        g->d_mjobs = reinterpret_cast<TkMJob*>(p + b0 + b1 + b2);
        if (b3) SDRX_HIP(hipMemcpy(g->d_mjobs, g->mjobs.data(), b3, hipMemcpyHostToDevice));
    }
    return SDRX_OK;
}

static int configure_channel(sdrx_chan_bank* b, int c, int32_t req_rate, int32_t req_fc)
{
    Channel& ch = b->ch[c];
    ch.req_rate = req_rate; ch.req_fc = req_fc;
    ch.n = plan_chain(b->in_rate, req_rate, req_fc, ch.modes, MAX_STAGES, &ch.out_rate, &ch.ofs);
    ch.passthrough = ch.n == 0;
    return SDRX_OK;
}

static int new_group(sdrx_chan_bank* b, const std::vector<int>& chans)
{
    if (chans.empty()) return SDRX_OK;
    Group* g = new (std::nothrow) Group;
    if (!g) return SDRX_ENOMEM;
    g->chans = chans;
    g->index = (int)b->groups.size();
    int rc = plan_group(b, g);
    if (rc) { free_group(g); return rc; }
    for (int c : chans) b->ch[c].group = (int)b->groups.size();
    b->groups.push_back(g);
    return SDRX_OK;
}

// A group none of whose channels is live any more (every one reconfigured away or removed) is dropped: its kernels,
// table uploads and device buffers would otherwise run / stay until reset.  Pending work on the stream may still read
// the group's buffers, hence the synchronisation.  Groups behind it move down one slot.
static int retire_dead_groups(sdrx_chan_bank* b)
{
    for (size_t gi = 0; gi < b->groups.size();) {
        Group* g = b->groups[gi];
        bool live = false;
        for (int c : g->chans) if (b->ch[(size_t)c].group == g->index) { live = true; break; }
        if (live) { gi++; continue; }
        SDRX_HIP(hipStreamSynchronize(b->stream));
        free_group(g);
        b->groups.erase(b->groups.begin() + (long)gi);
        for (size_t k = gi; k < b->groups.size(); k++) {
            const int old = b->groups[k]->index;
            b->groups[k]->index = (int)k;
            for (int c : b->groups[k]->chans) if (b->ch[(size_t)c].group == old) b->ch[(size_t)c].group = (int)k;
        }
    }
    return SDRX_OK;
}

// keep `used` bytes when growing a channel queue
static int grow_keep(sdrx_chan_bank* b, DevBuf& buf, size_t used, size_t need)
{
    if (need <= buf.cap) return SDRX_OK;
    size_t want = buf.cap ? buf.cap : (1 << 16);
    while (want < need) want *= 2;
    void* np = nullptr;
    SDRX_HIP(hipMalloc(&np, want));
    if (used) SDRX_HIP(hipMemcpyAsync(np, buf.p, used, hipMemcpyDeviceToDevice, b->stream));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    if (buf.p) (void)hipFree(buf.p);
    buf.p = np; buf.cap = want;
    return SDRX_OK;
}

static int dyn_slot(sdrx_chan_bank* b, size_t bytes, int* slot)
{
    const int s = b->dyn_next; b->dyn_next = (b->dyn_next + 1) % sdrx_chan_bank::RING;
    if (b->dyn_ev[s]) SDRX_HIP(hipEventSynchronize(b->dyn_ev[s]));
    else SDRX_HIP(hipEventCreateWithFlags(&b->dyn_ev[s], hipEventDisableTiming));
    if (bytes > b->dyn_cap[s]) {
        size_t want = b->dyn_cap[s] ? b->dyn_cap[s] : 4096; while (want < bytes) want *= 2;
        if (b->h_dyn[s]) (void)hipHostFree(b->h_dyn[s]);
        if (b->d_dyn[s]) (void)hipFree(b->d_dyn[s]);
        b->h_dyn[s] = b->d_dyn[s] = nullptr; b->dyn_cap[s] = 0;
        SDRX_HIP(hipHostMalloc(&b->h_dyn[s], want, hipHostMallocDefault));
        SDRX_HIP(hipMalloc(&b->d_dyn[s], want));
        b->dyn_cap[s] = want;
    }
    *slot = s;
    return SDRX_OK;
}

static int feed_group(sdrx_chan_bank* b, Group* g, const uint32_t* d_in, int64_t n)
{
    const int64_t T0 = g->T, T1 = g->T + n;
    // --- make room in the channel queues and the node-stream buffers
    for (int c : g->chans) {
        Channel& ch = b->ch[c];
        if (ch.group != g->index) continue;                // reconfigured away from this group
        const int64_t add = (T1 >> ch.n) - (T0 >> ch.n);
        int rc = grow_keep(b, ch.out, (size_t)ch.avail * 4, (size_t)(ch.avail + add) * 4 + 64); if (rc) return rc;
    }
    for (size_t si = 1; si < g->streams.size(); si++) {
        Stream& s = g->streams[si];
        const int64_t add = (T1 >> s.depth) - (T0 >> s.depth);
        int rc = s.mid.reserve((size_t)add * 4 + 64); if (rc) return rc;
    }
    // --- dynamic tables: [TkStream x n_streams][TkSink x n_sinks][TkHistJob x n_streams]
    const size_t ns = g->streams.size(), nk = g->sinks.size();
    const size_t o_sinks = ns * sizeof(TkStream), o_hist = o_sinks + nk * sizeof(TkSink), total = o_hist + ns * sizeof(TkHistJob);
    int slot; int rc = dyn_slot(b, total, &slot); if (rc) return rc;
    char* hp = static_cast<char*>(b->h_dyn[slot]); char* dp = static_cast<char*>(b->d_dyn[slot]);
    TkStream* hs = reinterpret_cast<TkStream*>(hp);
    TkSink* hk = reinterpret_cast<TkSink*>(hp + o_sinks);
    TkHistJob* hh = reinterpret_cast<TkHistJob*>(hp + o_hist);

    std::vector<long> segs(ns, 0);
    for (size_t si = 0; si < ns; si++) {
        Stream& s = g->streams[si];
        TkStream& t = hs[si];
        t.hist = s.hist[s.cur];
        t.in = si == 0 ? d_in : static_cast<const uint32_t*>(s.mid.p);
        t.t_old = T0 >> s.depth; t.t_new = T1 >> s.depth;
        t.subtree = s.subtree;
        t.c_first = t.t_old / TK_CHUNK;
        t.c_last = t.t_new > t.t_old ? (t.t_new - 1) / TK_CHUNK : t.c_first - 1;
        t.cps = 1;
        t.hist_len = s.hist_len;
        hh[si] = TkHistJob{ s.hist[s.cur], t.in, s.hist[s.cur ^ 1], t.t_new - t.t_old, s.hist_len };
    }
    for (size_t k = 0; k < nk; k++) {
        const SinkInfo& si = g->sinks[k];
        TkSink& t = hk[k];
        t.shift = 0; t.next = si.next;
        if (si.kind == 0) {
            Channel& ch = b->ch[si.ch];
            t.lo = T0 >> si.depth; t.hi = T1 >> si.depth;
            // element 0 of the queue's free space <-> absolute index lo; the address of index 0 is only ever used with an offset back into the buffer
            t.ptr0 = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(ch.out.p) - (uintptr_t)(4 * (t.lo - ch.avail)));
            t.shift = si.depth;
            if (ch.group != g->index) t.hi = t.lo;         // reconfigured away: still evaluated, not stored
        } else {
            Stream& s = g->streams[si.stream];
            t.lo = T0 >> s.depth; t.hi = T1 >> s.depth;
            t.ptr0 = reinterpret_cast<uint32_t*>(reinterpret_cast<uintptr_t>(s.mid.p) - (uintptr_t)(4 * t.lo));
        }
    }
    // chunks per segment, per pass: ~4 workgroups per CU overall, warm-up overhead <= 1/cps
    for (size_t p = 0; p < g->passes.size(); p++) {
        long total_chunks = 0;
        for (int si : g->passes[p]) total_chunks += std::max(0L, hs[si].c_last - hs[si].c_first + 1);
        // one round of workgroups (4 per CU) where the feed allows it: rounding DOWN here leaves a few segments for a second,
        // nearly empty round (61.44 M samples: 1072 workgroups for 1024 slots)
        long cps = (total_chunks + (long)b->cus * 4 - 1) / ((long)b->cus * 4);
        const char* env = getenv("SDRX_CHAN_CPS");
        if (env && atoi(env) > 0) cps = atoi(env);
        cps = std::max(1L, std::min(cps, 256L));
        for (int si : g->passes[p]) {
            hs[si].cps = (int)cps;
            segs[si] = (std::max(0L, hs[si].c_last - hs[si].c_first + 1) + cps - 1) / cps;
        }
    }
    SDRX_HIP(hipMemcpyAsync(dp, hp, total, hipMemcpyHostToDevice, b->stream));
    SDRX_HIP(hipEventRecord(b->dyn_ev[slot], b->stream));

    const TkStream* d_streams = reinterpret_cast<const TkStream*>(dp);
    const TkSink* d_sinks = reinterpret_cast<const TkSink*>(dp + o_sinks);
    const TkHistJob* d_hist = reinterpret_cast<const TkHistJob*>(dp + o_hist);
    rc = b->timer.begin(b->stream); if (rc) return rc;
    for (size_t p = 0; p < g->passes.size(); p++) {
        // the streams of one pass are contiguous in creation order; launch them as grid.y
        const std::vector<int>& ps = g->passes[p];
        if (ps.empty()) continue;
        long max_segs = 0;
        for (int si : ps) max_segs = std::max(max_segs, segs[si]);
        if (max_segs == 0) continue;
        const int s0 = ps.front(), cnt = (int)ps.size();
        size_t lds_bytes = 0;                              // per pass: a deep pass must not cost the shallow ones their occupancy
        for (int si : ps) lds_bytes = std::max(lds_bytes, (size_t)g->subtrees[(size_t)g->streams[(size_t)si].subtree].lds_dwords * 4);
        { const char* e = getenv("SDRX_CHAN_LDS_PAD_KB"); if (e) lds_bytes += (size_t)atoi(e) * 1024; }   // occupancy experiments only
        if (g->mfma)
            hipLaunchKernelGGL(tree_kernel<true>, dim3((unsigned)max_segs, (unsigned)cnt), dim3(TK_THREADS), lds_bytes, b->stream,
                               g->d_subtrees, g->d_nodes, g->d_arrays, d_streams + s0, d_sinks, g->d_mjobs);
        else
            hipLaunchKernelGGL(tree_kernel<false>, dim3((unsigned)max_segs, (unsigned)cnt), dim3(TK_THREADS), lds_bytes, b->stream,
                               g->d_subtrees, g->d_nodes, g->d_arrays, d_streams + s0, d_sinks, g->d_mjobs);
        SDRX_HIP(hipGetLastError());
        if (p == 0) {
            snprintf(b->last_name, sizeof b->last_name, g->mfma ? "tree_kernel<mfma>" : "tree_kernel<valu>");
            b->last_grid = (int)(max_segs * cnt); b->last_block = TK_THREADS; b->last_lds = (int)lds_bytes;
        }
    }
    rc = b->timer.end(b->stream); if (rc) return rc;
    long max_hist = TK_HIST;
    for (auto& s : g->streams) max_hist = std::max(max_hist, s.hist_len);
    hipLaunchKernelGGL(tree_hist_kernel, dim3((unsigned)(max_hist / 256), (unsigned)ns), dim3(256), 0, b->stream, d_hist);
    SDRX_HIP(hipGetLastError());
    for (auto& s : g->streams) s.cur ^= 1;
    for (int c : g->chans) {
        Channel& ch = b->ch[c];
        if (ch.group != g->index) continue;
        const int64_t add = (T1 >> ch.n) - (T0 >> ch.n);
        ch.last_off = ch.avail; ch.last_n = add; ch.avail += add;
    }
    g->T = T1;
    return SDRX_OK;
}

static int feed_passthrough(sdrx_chan_bank* b, const uint32_t* d_in, int64_t n)
{
    // no stage at all: DownChannelizer::feed hands the input straight to the sink (downchannelizer.cpp:57-60)
    for (auto& ch : b->ch) {
        if (!ch.passthrough) continue;
        int rc = grow_keep(b, ch.out, (size_t)ch.avail * 4, (size_t)(ch.avail + n) * 4 + 64); if (rc) return rc;
        SDRX_HIP(hipMemcpyAsync(static_cast<uint32_t*>(ch.out.p) + ch.avail, d_in, (size_t)n * 4, hipMemcpyDeviceToDevice, b->stream));
        ch.last_off = ch.avail; ch.last_n = n; ch.avail += n;
    }
    return SDRX_OK;
}

extern "C" {

int sdrx_chan_plan(int32_t in_rate, int32_t req_rate, int32_t req_fc, uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs)
{
    if (!modes || !out_rate || !residual_ofs) { set_error("sdrx_chan_plan: null argument"); return SDRX_EINVAL; }
    return plan_chain(in_rate, req_rate, req_fc, modes, MAX_STAGES, out_rate, residual_ofs);
}

int sdrx_chan_bank_create(sdrx_chan_bank_t** out, int device, int32_t in_rate, int32_t n_ch,
                          const int32_t* req_rate, const int32_t* req_fc)
{
    if (!out) { set_error("sdrx_chan_bank_create: null out"); return SDRX_EINVAL; }
    *out = nullptr;
    if (n_ch <= 0 || !req_rate || !req_fc || in_rate <= 0) { set_error("sdrx_chan_bank_create: bad argument"); return SDRX_EINVAL; }
    int rc = check_device(device); if (rc) return rc;
    SDRX_HIP(hipSetDevice(device));
    sdrx_chan_bank* b = new (std::nothrow) sdrx_chan_bank;
    if (!b) return SDRX_ENOMEM;
    b->device = device; b->in_rate = in_rate; b->cus = device_cu_count(device);
    hipError_t e = hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete b; return hip_fail(e, "hipStreamCreate", __FILE__, __LINE__); }
    b->stream = b->own_stream;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tree_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tree_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) { sdrx_chan_bank_destroy(b); return hip_fail(e, "hipFuncSetAttribute", __FILE__, __LINE__); }
    b->ch.resize((size_t)n_ch);
    std::vector<int> all;
    for (int c = 0; c < n_ch; c++) {
        configure_channel(b, c, req_rate[c], req_fc[c]);
        if (!b->ch[c].passthrough) all.push_back(c);
    }
    rc = new_group(b, all);
    if (rc) { sdrx_chan_bank_destroy(b); return rc; }
    SDRX_HIP(hipStreamSynchronize(b->stream));
    *out = b;
    return SDRX_OK;
}

int sdrx_chan_bank_destroy(sdrx_chan_bank_t* b)
{
    if (!b) return SDRX_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (Group* g : b->groups) free_group(g);
    for (auto& c : b->ch) c.out.release();
    b->stage_in.release(); b->scratch.release(); b->timer.release();
    for (int i = 0; i < sdrx_chan_bank::RING; i++) {
        if (b->h_dyn[i]) (void)hipHostFree(b->h_dyn[i]);
        if (b->d_dyn[i]) (void)hipFree(b->d_dyn[i]);
        if (b->dyn_ev[i]) (void)hipEventDestroy(b->dyn_ev[i]);
    }
    if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
    delete b;
    return SDRX_OK;
}

int sdrx_chan_bank_info(const sdrx_chan_bank_t* b, int32_t c, int32_t* n_stages, uint8_t* modes, int32_t* out_rate, int32_t* residual_ofs)
{
    if (!b || c < 0 || c >= (int32_t)b->ch.size()) { set_error("sdrx_chan_bank_info: bad channel"); return SDRX_EINVAL; }
    const Channel& ch = b->ch[(size_t)c];
    if (n_stages) *n_stages = ch.n;
    if (modes) memcpy(modes, ch.modes, (size_t)ch.n);
    if (out_rate) *out_rate = ch.out_rate;
    if (residual_ofs) *residual_ofs = ch.ofs;
    return SDRX_OK;
}

int sdrx_chan_bank_reconfigure(sdrx_chan_bank_t* b, int32_t c, int32_t req_rate, int32_t req_fc)
{
    if (!b || c < 0 || c >= (int32_t)b->ch.size()) { set_error("sdrx_chan_bank_reconfigure: bad channel"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    Channel& ch = b->ch[(size_t)c];
    // the old chain keeps being evaluated inside its group (its prefixes are shared) but stops
    // storing; the new chain starts from zero history in a group of its own
    // (freeFilterChain + createFilterChain, downchannelizer.cpp:167-171)
    ch.group = -1; ch.dead = false;                        // (a removed channel comes back with the new configuration)
    configure_channel(b, c, req_rate, req_fc);
    // a group whose last live channel this was (typically the single-channel group of an earlier reconfigure) goes away;
    // dead chains inside a group that still serves others keep being evaluated (shared prefixes) until the next reset
    int rc = retire_dead_groups(b); if (rc) return rc;
    if (ch.passthrough) return SDRX_OK;
    return new_group(b, std::vector<int>{ (int)c });
}

int sdrx_chan_bank_add_channel(sdrx_chan_bank_t* b, int32_t req_rate, int32_t req_fc, int32_t* channel)
{
    if (!b) { set_error("sdrx_chan_bank_add_channel: null bank"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    // a new DownChannelizer next to the existing ones: they keep their histories and queued output
    b->ch.emplace_back();
    const int c = (int)b->ch.size() - 1;
    configure_channel(b, c, req_rate, req_fc);
    if (channel) *channel = c;
    if (b->ch[(size_t)c].passthrough) return SDRX_OK;
    const int rc = new_group(b, std::vector<int>{ c });
    if (rc) { b->ch.pop_back(); if (channel) *channel = -1; }
    return rc;
}

int sdrx_chan_bank_remove_channel(sdrx_chan_bank_t* b, int32_t c)
{
    if (!b || c < 0 || c >= (int32_t)b->ch.size()) { set_error("sdrx_chan_bank_remove_channel: bad channel"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    // the index stays reserved (other channels keep theirs); the chain stops producing and its queue is dropped
    Channel& ch = b->ch[(size_t)c];
    ch.group = -1; ch.passthrough = false; ch.dead = true; ch.n = 0; ch.out_rate = 0; ch.ofs = 0;
    ch.avail = 0; ch.last_off = 0; ch.last_n = 0;
    return retire_dead_groups(b);
}

/* checkpoint of the filter state: per group its sample count and every stream's history (the raw stream and the node
 * streams; the reference's ring buffers are a pure function of them).  Layout: [magic, n_groups] then per group
 * [T, n_streams] and per stream [hist_len, hist_len x 4 bytes].  Queued, unread output is NOT part of it.  A state only
 * fits a bank with the same channels configured in the same order (the same plan): set_state checks the shape. */
static const int64_t CHAN_STATE_MAGIC = 0x7364727863686b31LL;      // "sdrxchk1"

int64_t sdrx_chan_bank_state_bytes(const sdrx_chan_bank_t* b)
{
    if (!b) return SDRX_EINVAL;
    int64_t n = 16;
    for (const Group* g : b->groups) { n += 16; for (const Stream& s : g->streams) n += 8 + (int64_t)s.hist_len * 4; }
    return n;
}

int sdrx_chan_bank_get_state(sdrx_chan_bank_t* b, void* host_buf)
{
    if (!b || !host_buf) { set_error("sdrx_chan_bank_get_state: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    char* p = static_cast<char*>(host_buf);
    auto put = [&](int64_t v) { std::memcpy(p, &v, 8); p += 8; };
    put(CHAN_STATE_MAGIC); put((int64_t)b->groups.size());
    for (Group* g : b->groups) {
        put(g->T); put((int64_t)g->streams.size());
        for (Stream& s : g->streams) {
            put(s.hist_len);
            SDRX_HIP(hipMemcpyAsync(p, s.hist[s.cur], (size_t)s.hist_len * 4, hipMemcpyDeviceToHost, b->stream));
            p += (size_t)s.hist_len * 4;
        }
    }
    SDRX_HIP(hipStreamSynchronize(b->stream));
    return SDRX_OK;
}

int sdrx_chan_bank_set_state(sdrx_chan_bank_t* b, const void* host_buf)
{
    if (!b || !host_buf) { set_error("sdrx_chan_bank_set_state: null argument"); return SDRX_EINVAL; }
    SDRX_HIP(hipSetDevice(b->device));
    const char* p = static_cast<const char*>(host_buf);
    auto get = [&]() { int64_t v; std::memcpy(&v, p, 8); p += 8; return v; };
    // first pass: the shape must be this bank's
    const char* q = p;
    bool ok = get() == CHAN_STATE_MAGIC && get() == (int64_t)b->groups.size();
    for (size_t gi = 0; ok && gi < b->groups.size(); gi++) {
        const Group* g = b->groups[gi];
        (void)get();
        ok = get() == (int64_t)g->streams.size();
        for (size_t si = 0; ok && si < g->streams.size(); si++) { ok = get() == g->streams[si].hist_len; p += (size_t)g->streams[si].hist_len * 4; }
    }
    if (!ok) { set_error("sdrx_chan_bank_set_state: the state was taken from a bank with another configuration"); return SDRX_EINVAL; }
    p = q; (void)get(); (void)get();
    SDRX_HIP(hipStreamSynchronize(b->stream));
    for (Group* g : b->groups) {
        g->T = get(); (void)get();
        for (Stream& s : g->streams) {
            (void)get();
            SDRX_HIP(hipMemcpyAsync(s.hist[s.cur], p, (size_t)s.hist_len * 4, hipMemcpyHostToDevice, b->stream));
            p += (size_t)s.hist_len * 4;
        }
    }
    SDRX_HIP(hipStreamSynchronize(b->stream));
    for (auto& ch : b->ch) { ch.avail = 0; ch.last_off = 0; ch.last_n = 0; }     // the queues belong to the old timeline
    return SDRX_OK;
}

int32_t sdrx_chan_bank_group_count(const sdrx_chan_bank_t* b) { return b ? (int32_t)b->groups.size() : SDRX_EINVAL; }

int sdrx_chan_bank_reset(sdrx_chan_bank_t* b)
{
    if (!b) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(b->device));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    for (Group* g : b->groups) free_group(g);
    b->groups.clear();
    std::vector<int> all;
    for (size_t c = 0; c < b->ch.size(); c++) {
        b->ch[c].avail = 0; b->ch[c].last_n = 0; b->ch[c].group = -1;
        if (!b->ch[c].passthrough && !b->ch[c].dead) all.push_back((int)c);
    }
    return new_group(b, all);
}

int sdrx_chan_bank_feed_dev(sdrx_chan_bank_t* b, const int16_t* d_iq, int64_t n_cplx)
{
    if (!b || n_cplx < 0 || (n_cplx > 0 && !d_iq)) { set_error("sdrx_chan_bank_feed_dev: bad argument"); return SDRX_EINVAL; }
    if (reinterpret_cast<uintptr_t>(d_iq) & 3u) { set_error("sdrx_chan_bank_feed_dev: d_iq must be 4-byte aligned"); return SDRX_EINVAL; }
    if (n_cplx == 0) return SDRX_OK;
    SDRX_HIP(hipSetDevice(b->device));
    const uint32_t* in = reinterpret_cast<const uint32_t*>(d_iq);
    int rc = feed_passthrough(b, in, n_cplx); if (rc) return rc;
    for (Group* g : b->groups) { rc = feed_group(b, g, in, n_cplx); if (rc) return rc; }
    return SDRX_OK;
}

int sdrx_chan_bank_feed(sdrx_chan_bank_t* b, const int16_t* iq, int64_t n_cplx)
{
    if (!b || n_cplx < 0 || (n_cplx > 0 && !iq)) { set_error("sdrx_chan_bank_feed: bad argument"); return SDRX_EINVAL; }
    if (n_cplx == 0) return SDRX_OK;
    SDRX_HIP(hipSetDevice(b->device));
    SDRX_HIP(hipStreamSynchronize(b->stream));            // staging buffer may still be read by the previous feed
    int rc = b->stage_in.reserve((size_t)n_cplx * 4); if (rc) return rc;
    SDRX_HIP(hipMemcpyAsync(b->stage_in.p, iq, (size_t)n_cplx * 4, hipMemcpyHostToDevice, b->stream));
    return sdrx_chan_bank_feed_dev(b, static_cast<const int16_t*>(b->stage_in.p), n_cplx);
}

int64_t sdrx_chan_bank_available(sdrx_chan_bank_t* b, int32_t c)
{
    if (!b || c < 0 || c >= (int32_t)b->ch.size()) return SDRX_EINVAL;
    return b->ch[(size_t)c].avail;
}

int64_t sdrx_chan_bank_read(sdrx_chan_bank_t* b, int32_t c, int16_t* out_iq, int64_t cap)
{
    if (!b || c < 0 || c >= (int32_t)b->ch.size() || cap < 0 || (cap > 0 && !out_iq)) { set_error("sdrx_chan_bank_read: bad argument"); return SDRX_EINVAL; }
    if (hipSetDevice(b->device) != hipSuccess) return SDRX_EHIP;
    Channel& ch = b->ch[(size_t)c];
    const int64_t n = std::min(cap, ch.avail);
    if (n == 0) return 0;
    hipError_t e = hipMemcpyAsync(out_iq, ch.out.p, (size_t)n * 4, hipMemcpyDeviceToHost, b->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(read)", __FILE__, __LINE__);
    const int64_t rest = ch.avail - n;
    if (rest > 0) {                                        // partial read: compact the queue
        int rc = b->scratch.reserve((size_t)rest * 4); if (rc) return rc;
        e = hipMemcpyAsync(b->scratch.p, static_cast<uint32_t*>(ch.out.p) + n, (size_t)rest * 4, hipMemcpyDeviceToDevice, b->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(ch.out.p, b->scratch.p, (size_t)rest * 4, hipMemcpyDeviceToDevice, b->stream);
        if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(compact)", __FILE__, __LINE__);
    }
    e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return hip_fail(e, "hipStreamSynchronize", __FILE__, __LINE__);
    ch.avail = rest; ch.last_off = 0; ch.last_n = 0;
    return n;
}

int64_t sdrx_chan_bank_skip(sdrx_chan_bank_t* b, int32_t c, int64_t n)
{
    if (!b || c < 0 || c >= (int32_t)b->ch.size()) { set_error("sdrx_chan_bank_skip: bad argument"); return SDRX_EINVAL; }
    Channel& ch = b->ch[(size_t)c];
    if (n < 0 || n >= ch.avail) { const int64_t k = ch.avail; ch.avail = 0; ch.last_off = 0; ch.last_n = 0; return k; }
    if (n == 0) return 0;
    if (hipSetDevice(b->device) != hipSuccess) return SDRX_EHIP;
    const int64_t rest = ch.avail - n;
    int rc = b->scratch.reserve((size_t)rest * 4); if (rc) return rc;
    hipError_t e = hipMemcpyAsync(b->scratch.p, static_cast<uint32_t*>(ch.out.p) + n, (size_t)rest * 4, hipMemcpyDeviceToDevice, b->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ch.out.p, b->scratch.p, (size_t)rest * 4, hipMemcpyDeviceToDevice, b->stream);
    if (e != hipSuccess) return hip_fail(e, "hipMemcpyAsync(skip)", __FILE__, __LINE__);
    ch.avail = rest; ch.last_off = 0; ch.last_n = 0;
    return n;
}

int sdrx_chan_bank_last_dev(sdrx_chan_bank_t* b, int32_t c, const int16_t** d_out_iq, int64_t* n_cplx)
{
    if (!b || c < 0 || c >= (int32_t)b->ch.size() || !d_out_iq || !n_cplx) { set_error("sdrx_chan_bank_last_dev: bad argument"); return SDRX_EINVAL; }
    Channel& ch = b->ch[(size_t)c];
    *d_out_iq = reinterpret_cast<const int16_t*>(static_cast<uint32_t*>(ch.out.p) + ch.last_off);
    *n_cplx = ch.last_n;
    return SDRX_OK;
}

int sdrx_chan_bank_sync(sdrx_chan_bank_t* b)
{
    if (!b) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(b->device));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    return SDRX_OK;
}

int sdrx_chan_bank_get_stream(sdrx_chan_bank_t* b, void** hip_stream)
{
    if (!b || !hip_stream) { set_error("sdrx_chan_bank_get_stream: null argument"); return SDRX_EINVAL; }
    *hip_stream = b->stream;
    return SDRX_OK;
}

int sdrx_chan_bank_set_stream(sdrx_chan_bank_t* b, void* hip_stream)
{
    if (!b) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(b->device));
    SDRX_HIP(hipStreamSynchronize(b->stream));
    b->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : b->own_stream;
    return SDRX_OK;
}

int sdrx_chan_bank_set_timing(sdrx_chan_bank_t* b, int enabled)
{
    if (!b) return SDRX_EINVAL;
    b->timer.enabled = enabled != 0;
    return SDRX_OK;
}

int sdrx_chan_bank_get_timing(sdrx_chan_bank_t* b, double* total_ms, int64_t* feeds, int reset)
{
    if (!b) return SDRX_EINVAL;
    SDRX_HIP(hipSetDevice(b->device));
    int rc = b->timer.collect(b->stream); if (rc) return rc;
    if (total_ms) *total_ms = b->timer.total_ms;
    if (feeds) *feeds = b->timer.count;
    if (reset) { b->timer.total_ms = 0; b->timer.count = 0; }
    return SDRX_OK;
}

int sdrx_chan_bank_last_launch(const sdrx_chan_bank_t* b, char* kernel_name, int name_cap, int* grid, int* block, int* lds_bytes)
{
    if (!b) return SDRX_EINVAL;
    if (kernel_name && name_cap > 0) snprintf(kernel_name, (size_t)name_cap, "%s", b->last_name);
    if (grid) *grid = b->last_grid;
    if (block) *block = b->last_block;
    if (lds_bytes) *lds_bytes = b->last_lds;
    return SDRX_OK;
}

} // extern "C"
