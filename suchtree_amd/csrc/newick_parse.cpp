// newick_parse.cpp -- native Newick ingest (host only, no GPU).
//
// Same observable result as suchtree_amd/newick.py::flat_tree_from_newick, which restates
// what the reference gets from dendropy (/root/reference/SuchTree/MuchTree.pyx:138-228):
// first tree of the text, [comments] dropped, quoted labels, polytomies resolved by joining
// the first two children under a new zero-length node appended last, in-order node ids,
// missing / zero lengths -> epsilon, numeric internal labels -> support.
//
// This is an accelerator, not a second source of truth: on any syntax error or any token
// whose Python meaning it is not sure to reproduce it reports "unsupported" and the caller
// falls back to the Python implementation, which owns the error behaviour.
#include <cerrno>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/suchtree_hip.h"

namespace {

struct PNode {
    int32_t parent = -1;
    int32_t first = -1, last = -1, next = -1;   // ordered child list
    int32_t n_child = 0;
    int64_t label_off = -1;   // into pool
    int32_t label_len = 0;
    double length = 0.0;
    bool has_length = false;
};

struct Parsed {
    int64_t n = 0, n_leaves = 0;
    int32_t root = -1, depth = 0;
    std::vector<int32_t> parent, left, right;
    std::vector<float> support, distance;
    std::vector<int32_t> leaf_ids;
    std::string names;                 // leaf names, concatenated in in-order order
    std::vector<int64_t> name_off;     // n_leaves + 1
};

bool is_plain_number(const char *s, int len)
{
    if (len <= 0 || len > 60) return false;
    bool digit = false;
    for (int i = 0; i < len; i++) {
        const char c = s[i];
        if (c >= '0' && c <= '9') digit = true;
        else if (c != '+' && c != '-' && c != '.' && c != 'e' && c != 'E') return false;
    }
    return digit;
}

// 1 = parsed, 0 = not a number in C or Python, -1 = unsure (let Python's float() decide)
int parse_number(const char *s, int len, double *out)
{
    if (!is_plain_number(s, len)) {
        if (len <= 0) return 0;
        bool only_numeric_chars = true;
        for (int i = 0; i < len; i++) {
            const char c = s[i];
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') return -1;   // float() strips
            if (!((c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E' || c == '_'))
                only_numeric_chars = false;
        }
        if (only_numeric_chars) return -1;   // underscores (1_000), over-long literals
        // inf / infinity / nan with an optional sign are floats for Python
        const int off = (s[0] == '+' || s[0] == '-') ? 1 : 0;
        char low[9] = {0};
        const int m = len - off;
        if (m >= 3 && m <= 8) {
            for (int i = 0; i < m; i++) low[i] = (char)(s[off + i] | 0x20);
            if (!std::strcmp(low, "inf") || !std::strcmp(low, "infinity") || !std::strcmp(low, "nan")) return -1;
        }
        return 0;
    }
    char buf[64];
    std::memcpy(buf, s, (size_t)len);
    buf[len] = 0;
    char *end = nullptr;
    errno = 0;
    const double v = std::strtod(buf, &end);
    if (end != buf + len) return 0;   // e.g. "1e" or "--1": float() raises too
    *out = v;
    return 1;
}

const double kEpsilon = DBL_EPSILON;   // np.finfo(np.float64).eps, MuchTree.pyx:136

// returns 0 ok, 1 unsupported (caller falls back to Python)
int parse(const char *text, int64_t len, Parsed &P)
{
    std::vector<PNode> nodes;
    std::string pool;
    nodes.reserve(1024);
    nodes.emplace_back();
    int32_t cur = 0;
    bool want_length = false, seen_any = false;

    auto new_child = [&](int32_t p) {
        nodes.emplace_back();
        const int32_t k = (int32_t)nodes.size() - 1;
        nodes[(size_t)k].parent = p;
        PNode &pp = nodes[(size_t)p];
        if (pp.last < 0) pp.first = k; else nodes[(size_t)pp.last].next = k;
        pp.last = k;
        pp.n_child++;
        return k;
    };

    int64_t i = 0;
    bool closed = false;
    while (i < len && !closed) {
        const unsigned char c = (unsigned char)text[i];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v') { i++; continue; }
        if (c >= 0x80) return 1;   // non-ASCII: leave str semantics to Python
        if (c == '[') {
            const void *e = std::memchr(text + i, ']', (size_t)(len - i));
            if (!e) return 1;
            i = (const char *)e - text + 1;
            continue;
        }
        int64_t tok_off = -1;
        int32_t tok_len = 0;
        bool is_label = false;
        if (c == '\'') {
            // quoted label, '' is an escaped quote
            const int64_t start = (int64_t)pool.size();
            int64_t j = i + 1;
            for (;;) {
                if (j >= len) return 1;
                if (text[j] == '\'') {
                    if (j + 1 < len && text[j + 1] == '\'') { pool.push_back('\''); j += 2; continue; }
                    break;
                }
                if ((unsigned char)text[j] >= 0x80) return 1;
                pool.push_back(text[j]);
                j++;
            }
            tok_off = start;
            tok_len = (int32_t)((int64_t)pool.size() - start);
            is_label = true;
            i = j + 1;
        } else if (c == '(' || c == ')' || c == ',' || c == ':' || c == ';') {
            i++;
            seen_any = true;
            if (want_length) return 1;
            if (c == '(') cur = new_child(cur);
            else if (c == ',') {
                if (nodes[(size_t)cur].parent < 0) return 1;
                cur = new_child(nodes[(size_t)cur].parent);
            } else if (c == ')') {
                if (nodes[(size_t)cur].parent < 0) return 1;
                cur = nodes[(size_t)cur].parent;
            } else if (c == ':') want_length = true;
            else closed = true;
            continue;
        } else if (c == ']') {
            return 1;
        } else {
            int64_t j = i;
            while (j < len) {
                const unsigned char d = (unsigned char)text[j];
                if (d == ' ' || d == '\t' || d == '\n' || d == '\r' || d == '\f' || d == '\v' || d == '(' ||
                    d == ')' || d == '[' || d == ']' || d == ',' || d == ':' || d == ';')
                    break;     // (a quote inside an unquoted token is an ordinary character, as in dendropy's tokenizer)
                if (d >= 0x80) return 1;
                j++;
            }
            const int64_t start = (int64_t)pool.size();
            pool.append(text + i, (size_t)(j - i));
            tok_off = start;
            tok_len = (int32_t)(j - i);
            is_label = true;
            i = j;
        }
        if (is_label) {
            seen_any = true;
            if (want_length) {
                double v;
                if (parse_number(pool.data() + tok_off, tok_len, &v) != 1) return 1;
                nodes[(size_t)cur].length = v;
                nodes[(size_t)cur].has_length = true;
                want_length = false;
            } else {
                nodes[(size_t)cur].label_off = tok_off;
                nodes[(size_t)cur].label_len = tok_len;
            }
        }
    }
    if (!seen_any || cur != 0 || want_length) return 1;

    // resolve polytomies: join the first two children under a new zero-length node, append it last
    const size_t n0 = nodes.size();
    for (size_t x = 0; x < n0; x++) {
        if (nodes[x].n_child <= 2) continue;
        std::deque<int32_t> ch;
        for (int32_t k = nodes[x].first; k >= 0; k = nodes[(size_t)k].next) ch.push_back(k);
        while (ch.size() > 2) {
            const int32_t c1 = ch[0], c2 = ch[1];
            ch.pop_front();
            ch.pop_front();
            nodes.emplace_back();
            const int32_t nn = (int32_t)nodes.size() - 1;
            PNode &N = nodes[(size_t)nn];
            N.parent = (int32_t)x;
            N.first = c1; N.last = c2; N.n_child = 2;
            N.length = 0.0; N.has_length = true;
            nodes[(size_t)c1].parent = nn; nodes[(size_t)c1].next = c2;
            nodes[(size_t)c2].parent = nn; nodes[(size_t)c2].next = -1;
            ch.push_back(nn);
        }
        nodes[x].first = ch[0];
        nodes[x].last = ch[1];
        nodes[(size_t)ch[0]].next = ch[1];
        nodes[(size_t)ch[1]].next = -1;
        nodes[x].n_child = 2;
    }

    // in-order numbering (iterative); any node with exactly one child is not traversable
    const int64_t n = (int64_t)nodes.size();
    if (n > INT32_MAX) return 1;
    std::vector<int32_t> id((size_t)n, -1);
    std::vector<int32_t> order;
    order.reserve((size_t)n);
    {
        std::vector<int32_t> stack;
        std::vector<uint8_t> state((size_t)n, 0);
        stack.push_back(0);
        while (!stack.empty()) {
            const int32_t x = stack.back();
            const PNode &N = nodes[(size_t)x];
            if (N.n_child == 0) {
                order.push_back(x);
                stack.pop_back();
            } else if (N.n_child == 2) {
                if (state[(size_t)x] == 0) { state[(size_t)x] = 1; stack.push_back(N.first); }
                else if (state[(size_t)x] == 1) { state[(size_t)x] = 2; order.push_back(x); stack.push_back(N.last); }
                else stack.pop_back();
            } else {
                return 1;
            }
        }
    }
    if ((int64_t)order.size() != n) return 1;
    for (int64_t k = 0; k < n; k++) id[(size_t)order[(size_t)k]] = (int32_t)k;

    P.n = n;
    P.parent.assign((size_t)n, -1);
    P.left.assign((size_t)n, -1);
    P.right.assign((size_t)n, -1);
    P.support.assign((size_t)n, -1.0f);
    P.distance.assign((size_t)n, -1.0f);
    P.name_off.push_back(0);
    std::vector<int32_t> depth((size_t)n, 0);
    for (int64_t k = 0; k < n; k++) {
        const int32_t x = order[(size_t)k];
        const PNode &N = nodes[(size_t)x];
        if (N.n_child == 0) {
            if (N.label_off < 0) return 1;   // leaf without a name
            P.leaf_ids.push_back((int32_t)k);
            P.names.append(pool.data() + N.label_off, (size_t)N.label_len);
            P.name_off.push_back((int64_t)P.names.size());
        } else {
            const int32_t l = id[(size_t)N.first], r = id[(size_t)N.last];
            P.left[(size_t)k] = l;
            P.right[(size_t)k] = r;
            P.parent[(size_t)l] = (int32_t)k;
            P.parent[(size_t)r] = (int32_t)k;
            if (N.label_off >= 0) {
                double v;
                const int rc = parse_number(pool.data() + N.label_off, N.label_len, &v);
                if (rc < 0) return 1;
                if (rc == 1) P.support[(size_t)k] = (float)v;
            }
        }
        if (x != 0) {
            const double e = (N.has_length && N.length != 0.0) ? N.length : kEpsilon;
            P.distance[(size_t)k] = (float)e;   // the reference stores C floats (pyx:60,215)
        }
    }
    P.root = id[0];
    P.n_leaves = (int64_t)P.leaf_ids.size();
    // leaf names must be unique for the name dict to have n_leaves entries: Python checks nothing,
    // later names overwrite earlier ones -- keep that behaviour by leaving it to the caller.
    // depth (nodes on the longest leaf->root path): parents precede children in `nodes` creation
    // order except for polytomy joins, so walk an explicit stack from the root.
    {
        std::vector<int32_t> stack{P.root};
        int32_t deepest = 0;
        while (!stack.empty()) {
            const int32_t k = stack.back();
            stack.pop_back();
            if (P.left[(size_t)k] < 0) { if (depth[(size_t)k] > deepest) deepest = depth[(size_t)k]; continue; }
            depth[(size_t)P.left[(size_t)k]] = depth[(size_t)k] + 1;
            depth[(size_t)P.right[(size_t)k]] = depth[(size_t)k] + 1;
            stack.push_back(P.left[(size_t)k]);
            stack.push_back(P.right[(size_t)k]);
        }
        P.depth = deepest + 1;
    }
    return 0;
}

}  // namespace

struct st_newick {
    Parsed P;
};

extern "C" {

int st_newick_open(const char *text, int64_t len, st_newick **out, int64_t *n_nodes, int64_t *n_leaves,
                   int64_t *names_bytes, int32_t *root, int32_t *depth)
{
    if (!text || len < 0 || !out) return ST_ERR_ARG;
    *out = nullptr;
    st_newick *h = new (std::nothrow) st_newick();
    if (!h) return ST_ERR_NOMEM;
    if (parse(text, len, h->P) != 0) {
        delete h;
        return ST_ERR_TREE;   // "unsupported": the caller falls back to the Python parser
    }
    *out = h;
    if (n_nodes) *n_nodes = h->P.n;
    if (n_leaves) *n_leaves = h->P.n_leaves;
    if (names_bytes) *names_bytes = (int64_t)h->P.names.size();
    if (root) *root = h->P.root;
    if (depth) *depth = h->P.depth;
    return ST_OK;
}

int st_newick_fill(const st_newick *h, int32_t *parent, int32_t *left, int32_t *right, float *support,
                   float *distance, int32_t *leaf_ids, char *names, int64_t *name_offsets)
{
    if (!h) return ST_ERR_ARG;
    const Parsed &P = h->P;
    const size_t n = (size_t)P.n;
    if (parent) std::memcpy(parent, P.parent.data(), n * 4);
    if (left) std::memcpy(left, P.left.data(), n * 4);
    if (right) std::memcpy(right, P.right.data(), n * 4);
    if (support) std::memcpy(support, P.support.data(), n * 4);
    if (distance) std::memcpy(distance, P.distance.data(), n * 4);
    if (leaf_ids) std::memcpy(leaf_ids, P.leaf_ids.data(), (size_t)P.n_leaves * 4);
    if (names && !P.names.empty()) std::memcpy(names, P.names.data(), P.names.size());
    if (name_offsets) std::memcpy(name_offsets, P.name_off.data(), ((size_t)P.n_leaves + 1) * 8);
    return ST_OK;
}

void st_newick_close(st_newick *h) { delete h; }

}  // extern "C"
