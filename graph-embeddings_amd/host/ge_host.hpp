// ge_host.hpp -- C++ host side above the C ABI, mirroring the reference's operator interfaces for the
// hot path (the reference is compiled Java; no JDK exists in the build image, so the host that a
// Java maintainer would write over JNI -- see INTEGRATION.md -- is restated here in C++).
//
//   Configuration / check / InvalidConfigurationException   J/util/config/Configuration.java, J/util/read/ConfigReader.java
//   CoOccurrenceMatrix, BookmarkColoring                    J/util/CoOccurrenceMatrix.java:6-17, J/bca/BookmarkColoring.java
//   IOptimizer, Adagrad, Optimum, CostFunction choice       J/opt/IOptimizer.java, J/opt/grad/Adagrad.java, J/opt/Optimizer.java:66-120
//   EmbeddingTextWriter                                     J/util/write/EmbeddingTextWriter.java
//   Main.createFileName / createOptimizer / runProgram      J/Main.java:29-131
//   N-Triples graph ingest with Rdf2GrphConverter's vertex/edge rules   J/convert/Rdf2GrphConverter.java:71-114,195-241
// (J/ = src/main/java/org/uu/nl/embedding/).  All numeric work goes through libgeglove.so.
#pragma once

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <filesystem>
#include <fstream>
#include <map>
#include <set>
#include <iterator>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/geglove.h"

namespace ge_host {

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
struct InvalidConfigurationException : std::runtime_error {     // Configuration.java:496-500
    explicit InvalidConfigurationException(const std::string &m) : std::runtime_error("Invalid configuration: " + m) {}
};
struct NativeError : std::runtime_error {
    ge_status status;
    NativeError(ge_status s, const std::string &what) : std::runtime_error(what), status(s) {}
};
inline void check(ge_status s) {
    if (s != GE_OK) throw NativeError(s, std::string("geglove error ") + std::to_string(s) + ": " + ge_last_error());
}

// ------------------------------------------------------------------------------------------------
// Java number -> string conversions used by the writer header and the file name
// ------------------------------------------------------------------------------------------------
// Shortest decimal digits that read back as `a` (as a float when max_prec == 8), correctly rounded: what Double.toString /
// Float.toString print.  std::to_chars gives exactly that (shortest round-trip, closest among the shortest) without the
// print-and-parse loop; n = number of digits written to dig[], value = d1.d2d3... * 10^exp10.
inline int shortest_digits_fast(double a, bool as_float, char *dig, int &exp10) {
    char buf[64];
    const std::to_chars_result r = as_float ? std::to_chars(buf, buf + sizeof buf, (float)a, std::chars_format::scientific)
                                            : std::to_chars(buf, buf + sizeof buf, a, std::chars_format::scientific);
    int n = 0;
    const char *q = buf;
    for (; q < r.ptr && *q != 'e'; ++q) if (*q >= '0' && *q <= '9') dig[n++] = *q;
    int ex = 0; bool neg = false;
    for (++q; q < r.ptr; ++q) { if (*q == '-') neg = true; else if (*q >= '0' && *q <= '9') ex = ex * 10 + (*q - '0'); }
    exp10 = neg ? -ex : ex;
    while (n > 1 && dig[n - 1] == '0') --n;
    return n;
}
inline std::string shortest_digits(double a, int max_prec, int &exp10) {   // digits d1d2..., value = 0.d1d2.. * 10^(exp10+1)
    char dig[32];
    const int n = shortest_digits_fast(a, max_prec == 8, dig, exp10);
    return std::string(dig, (size_t)n);
}
// Double.toString / Float.toString: decimal for 1e-3 <= |x| < 1e7, otherwise d.dddE[-]n
inline std::string java_number(double v, bool is_float) {
    if (v != v) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    if (v == 0) return std::signbit(v) ? "-0.0" : "0.0";
    int ex;
    const std::string d = shortest_digits(std::fabs(v), is_float ? 8 : 17, ex);
    std::string out = v < 0 ? "-" : "";
    if (ex >= -3 && ex < 7) {
        if (ex < 0) { out += "0."; out.append((size_t)(-ex - 1), '0'); out += d; }
        else {
            for (int k = 0; k <= ex; ++k) out.push_back(k < (int)d.size() ? d[(size_t)k] : '0');
            out.push_back('.');
            if ((int)d.size() > ex + 1) out += d.substr((size_t)ex + 1); else out.push_back('0');
        }
    } else {
        out.push_back(d[0]); out.push_back('.');
        if (d.size() > 1) out += d.substr(1); else out.push_back('0');
        out += "E" + std::to_string(ex);
    }
    return out;
}
// String.format("%11.6E", v): HALF_UP on the shortest repr digits (EmbeddingTextWriter.java:134).  Writes the field (at least
// 11 characters, right-aligned) to out and returns its length; out holds >= 32 characters.
inline int java_format_11_6E_to(double v, char *out) {
    if (v != v) { std::memcpy(out, "        NaN", 11); return 11; }
    if (std::isinf(v)) { const char *t = v > 0 ? "   Infinity" : "  -Infinity"; std::memcpy(out, t, 11); return 11; }
    int ex = 0; char d[32]; int n = 1;
    if (v == 0) d[0] = '0'; else n = shortest_digits_fast(std::fabs(v), false, d, ex);
    int dig[7];
    for (int k = 0; k < 7; ++k) dig[k] = k < n ? d[k] - '0' : 0;
    if (n > 7 && d[7] >= '5') {
        int k = 6;
        while (k >= 0) { if (++dig[k] < 10) break; dig[k] = 0; --k; }
        if (k < 0) { dig[0] = 1; for (int q = 1; q < 7; ++q) dig[q] = 0; ++ex; }
    }
    char body[24]; int m = 0;
    if (std::signbit(v)) body[m++] = '-';
    body[m++] = (char)('0' + dig[0]); body[m++] = '.';
    for (int k = 1; k < 7; ++k) body[m++] = (char)('0' + dig[k]);
    body[m++] = 'E'; body[m++] = ex < 0 ? '-' : '+';
    const int ax = ex < 0 ? -ex : ex;
    if (ax >= 100) body[m++] = (char)('0' + ax / 100);
    body[m++] = (char)('0' + (ax / 10) % 10); body[m++] = (char)('0' + ax % 10);
    int len = 0;
    for (int k = m; k < 11; ++k) out[len++] = ' ';
    std::memcpy(out + len, body, (size_t)m);
    return len + m;
}
inline std::string java_format_11_6E(double v) { char out[32]; const int n = java_format_11_6E_to(v, out); return std::string(out, (size_t)n); }

// ------------------------------------------------------------------------------------------------
// YAML subset: block maps, block lists ("- "), flow lists [a, b], scalars, comments.  That is all
// dblp/onstage/saa.config.yml use.  A '#' starts a comment only at line start or after whitespace
// (predicate URIs contain '#').
// ------------------------------------------------------------------------------------------------
struct YNode {
    enum Kind { Null, Scalar, Map, List } kind = Null;
    std::string scalar;
    std::vector<std::pair<std::string, YNode>> map;     // insertion order kept (SnakeYAML builds a LinkedHashMap)
    std::vector<YNode> list;
    const YNode *get(const std::string &k) const {
        for (auto &kv : map) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};
namespace detail {
inline std::string trim(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
    return s.substr(a, b - a);
}
inline std::string unquote(std::string s) {
    s = trim(s);
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}
inline std::string strip_comment(const std::string &l) {
    bool sq = false, dq = false;
    for (size_t k = 0; k < l.size(); ++k) {
        if (l[k] == '\'' && !dq) sq = !sq;
        else if (l[k] == '"' && !sq) dq = !dq;
        else if (l[k] == '#' && !sq && !dq && (k == 0 || l[k - 1] == ' ' || l[k - 1] == '\t')) return l.substr(0, k);
    }
    return l;
}
struct Line { int indent; std::string text; };
inline YNode parse_value_inline(const std::string &v) {
    YNode n;
    const std::string t = trim(v);
    if (!t.empty() && t.front() == '[' && t.back() == ']') {
        n.kind = YNode::List;
        std::string inner = t.substr(1, t.size() - 2), item;
        std::stringstream ss(inner);
        while (std::getline(ss, item, ',')) { item = unquote(item); if (!item.empty()) { YNode c; c.kind = YNode::Scalar; c.scalar = item; n.list.push_back(c); } }
        return n;
    }
    n.kind = YNode::Scalar; n.scalar = unquote(t);
    if (n.scalar == "~" || n.scalar == "null") { n.kind = YNode::Null; n.scalar.clear(); }
    return n;
}
inline bool split_key(const std::string &t, std::string &key, std::string &val) {
    size_t pos = std::string::npos;
    for (size_t k = 0; k < t.size(); ++k)
        if (t[k] == ':' && (k + 1 == t.size() || t[k + 1] == ' ' || t[k + 1] == '\t')) { pos = k; break; }
    if (pos == std::string::npos) return false;
    key = unquote(t.substr(0, pos)); val = trim(t.substr(pos + 1));
    return true;
}
inline YNode parse_block(const std::vector<Line> &L, size_t &i, int indent) {
    YNode n;
    if (i >= L.size()) return n;
    const bool is_list = L[i].text.rfind("- ", 0) == 0 || L[i].text == "-";
    n.kind = is_list ? YNode::List : YNode::Map;
    while (i < L.size() && L[i].indent == indent) {
        std::string t = L[i].text;
        if (is_list) {
            if (!(t.rfind("- ", 0) == 0 || t == "-")) break;
            std::string rest = t.size() > 2 ? trim(t.substr(2)) : "";
            std::string k, v;
            if (!rest.empty() && split_key(rest, k, v)) {
                // "- key: value" opens a map whose further keys are indented by indent+2
                std::vector<Line> sub;
                sub.push_back({indent + 2, rest});
                size_t j = i + 1;
                while (j < L.size() && L[j].indent > indent) { sub.push_back(L[j]); ++j; }
                size_t si = 0;
                n.list.push_back(parse_block(sub, si, indent + 2));
                i = j;
            } else { n.list.push_back(parse_value_inline(rest)); ++i; }
        } else {
            std::string k, v;
            if (!split_key(t, k, v)) throw InvalidConfigurationException("cannot parse line: " + t);
            ++i;
            if (!v.empty()) n.map.push_back({k, parse_value_inline(v)});
            else if (i < L.size() && L[i].indent > indent) { const int ci = L[i].indent; n.map.push_back({k, parse_block(L, i, ci)}); }
            else n.map.push_back({k, YNode()});
        }
    }
    return n;
}
}  // namespace detail
inline YNode parse_yaml(std::istream &in) {
    std::vector<detail::Line> L;
    std::string line;
    while (std::getline(in, line)) {
        line = detail::strip_comment(line);
        size_t a = 0; while (a < line.size() && line[a] == ' ') ++a;
        const std::string t = detail::trim(line);
        if (t.empty() || t == "---") continue;
        L.push_back({(int)a, t});
    }
    size_t i = 0;
    if (L.empty()) return YNode();
    return detail::parse_block(L, i, L[0].indent);
}

// ------------------------------------------------------------------------------------------------
// Configuration (tolerant bean)
// ------------------------------------------------------------------------------------------------
struct Configuration {
    std::string graph, method;
    int dim = 0, threads = 0;
    std::vector<std::pair<std::string, float>> weights;
    bool has_weights = false;
    std::vector<std::map<std::string, std::string>> similarity;
    struct { double alpha = 0, epsilon = 0; bool directed = false; std::string normalize; bool present = false; } bca;
    struct { std::string method = "adagrad"; double tolerance = 0; int maxiter = 0; } opt;
    struct { bool present = false; double variance = 0; } pca;
    struct Output { bool present = false; std::string name; bool has_uri = false, has_blank = false, has_predicate = false, has_literal = false;
                    std::vector<std::string> uri, blank, predicate, literal; } output;
    // new, optional block `device:` (never changes the meaning of a reference key)
    struct { std::string mode = "hogwild", shuffle = "device", hot = "auto", dtype = "f32", save_coo, load_coo;
             long long seed = 0; bool has_seed = false; int id = 0; int workers = 0;
             // Hogwild tuning (ge_glove_cfg: 0 = library default), layout: list of fixed_cuts | plain_long_rows | separate_tables | packed_records
             double hot_theta = 0, stale_budget = 0; int flush_every = 0, blocks_per_cu = 0, layout_flags = 0;
             long long bca_table_slots = 0, bca_pool_entries = 0;
             // multi-GPU: gpus ranks, one thread each, rows sharded, context exchanged through ge_sync
             int gpus = 1, accum_every = 0, hub_segments = 0; std::string exchange = "overlap", transport = "auto", wire = "bf16"; } device;
    std::vector<std::string> ignored_keys;     // legacy keys of the shipped YAMLs that the bean does not know

    int getThreads() const {       // Configuration.java:71-73
        if (threads != 0) return threads;
        const unsigned hc = std::thread::hardware_concurrency();
        return std::max(1, (int)hc - 1);
    }
    std::string getNormalize() const { return bca.normalize.empty() ? "none" : bca.normalize; }
    bool usingPca() const { return pca.present; }
    bool usingWeights() const { return has_weights && !weights.empty(); }
    bool usingSimilarity() const { return !similarity.empty(); }

    static double num(const YNode *n, double def = 0) { return (n && n->kind == YNode::Scalar) ? std::strtod(n->scalar.c_str(), nullptr) : def; }
    static std::string str(const YNode *n) { return (n && n->kind == YNode::Scalar) ? n->scalar : std::string(); }
    static bool boolean(const YNode *n) { const std::string s = str(n); return s == "true" || s == "True" || s == "yes" || s == "on"; }
    static void strlist(const YNode *n, bool &present, std::vector<std::string> &out) {
        if (!n) return;
        present = n->kind != YNode::Null;
        if (n->kind == YNode::List) for (auto &c : n->list) out.push_back(c.scalar);
        else if (n->kind == YNode::Scalar && !n->scalar.empty()) out.push_back(n->scalar);
        if (n->kind == YNode::Null) present = false;
    }

    static Configuration load(const std::string &path) {       // ConfigReader.load
        std::ifstream f(path);
        if (!f) throw std::runtime_error("Cannot find configuration file + " + path);
        const YNode root = parse_yaml(f);
        Configuration c;
        if (root.kind != YNode::Map) throw InvalidConfigurationException("top level must be a mapping");
        for (auto &kv : root.map) {
            const std::string &k = kv.first; const YNode &v = kv.second;
            if (k == "graph") c.graph = v.scalar;
            else if (k == "method") c.method = v.scalar;
            else if (k == "dim") c.dim = (int)num(&v);
            else if (k == "threads") c.threads = (int)num(&v);
            else if (k == "weights") { c.has_weights = true; for (auto &w : v.map) c.weights.push_back({w.first, (float)num(&w.second)}); }
            else if (k == "similarity") {
                for (auto &item : v.list) {
                    std::map<std::string, std::string> m;
                    for (auto &q : item.map) m[q.first] = q.second.scalar;
                    if (m.count("predicate")) {      // legacy key of the shipped YAMLs: source = target
                        if (!m.count("sourcePredicate")) m["sourcePredicate"] = m["predicate"];
                        if (!m.count("targetPredicate")) m["targetPredicate"] = m["predicate"];
                    }
                    c.similarity.push_back(m);
                }
            } else if (k == "bca") {
                c.bca.present = true;
                for (auto &q : v.map) {
                    if (q.first == "alpha") c.bca.alpha = num(&q.second);
                    else if (q.first == "epsilon") c.bca.epsilon = num(&q.second);
                    else if (q.first == "directed") c.bca.directed = boolean(&q.second);
                    else if (q.first == "normalize") c.bca.normalize = q.second.scalar;
                    else c.ignored_keys.push_back("bca." + q.first);           // bca.reverse, bca.predicates
                }
            } else if (k == "opt") {
                for (auto &q : v.map) {
                    if (q.first == "method") c.opt.method = q.second.scalar;
                    else if (q.first == "tolerance") c.opt.tolerance = num(&q.second);
                    else if (q.first == "maxiter") c.opt.maxiter = (int)num(&q.second);
                    else c.ignored_keys.push_back("opt." + q.first);
                }
            } else if (k == "pca") { c.pca.present = true; c.pca.variance = num(v.get("variance")); }
            else if (k == "output") {
                c.output.present = true;
                c.output.name = str(v.get("name"));
                strlist(v.get("uri"), c.output.has_uri, c.output.uri);
                strlist(v.get("blank"), c.output.has_blank, c.output.blank);
                strlist(v.get("predicate"), c.output.has_predicate, c.output.predicate);
                strlist(v.get("literal"), c.output.has_literal, c.output.literal);
                // an explicitly empty flow list `uri: []` counts as "present" (outputUriNodes() is `uri != null`)
                for (auto &q : v.map) if (q.second.kind == YNode::List && q.second.list.empty()) {
                    if (q.first == "uri") c.output.has_uri = true; else if (q.first == "blank") c.output.has_blank = true;
                    else if (q.first == "predicate") c.output.has_predicate = true; else if (q.first == "literal") c.output.has_literal = true;
                }
            } else if (k == "device") {
                for (auto &q : v.map) {
                    if (q.first == "mode") c.device.mode = q.second.scalar;
                    else if (q.first == "shuffle") c.device.shuffle = q.second.scalar;
                    else if (q.first == "hot") c.device.hot = q.second.scalar;
                    else if (q.first == "seed") { c.device.seed = std::strtoll(q.second.scalar.c_str(), nullptr, 10); c.device.has_seed = true; }
                    else if (q.first == "id") c.device.id = (int)num(&q.second);
                    else if (q.first == "workers") c.device.workers = (int)num(&q.second);
                    else if (q.first == "dtype") c.device.dtype = q.second.scalar;
                    else if (q.first == "save_coo") c.device.save_coo = q.second.scalar;      // COO checkpoint
                    else if (q.first == "load_coo") c.device.load_coo = q.second.scalar;
                    else if (q.first == "gpus") c.device.gpus = std::max(1, (int)num(&q.second));
                    else if (q.first == "exchange") c.device.exchange = q.second.scalar;        // overlap | sync
                    else if (q.first == "transport") c.device.transport = q.second.scalar;      // auto | rccl | host
                    else if (q.first == "wire") c.device.wire = q.second.scalar;                // bf16 | f32
                    else if (q.first == "accum_every") c.device.accum_every = (int)num(&q.second);
                    else if (q.first == "hub_segments") c.device.hub_segments = (int)num(&q.second);
                    else if (q.first == "hot_theta") c.device.hot_theta = num(&q.second);
                    else if (q.first == "stale_budget") c.device.stale_budget = num(&q.second);
                    else if (q.first == "flush_every") c.device.flush_every = (int)num(&q.second);
                    else if (q.first == "blocks_per_cu") c.device.blocks_per_cu = (int)num(&q.second);
                    else if (q.first == "bca_table_slots") c.device.bca_table_slots = (long long)num(&q.second);
                    else if (q.first == "bca_pool_entries") c.device.bca_pool_entries = (long long)num(&q.second);
                    else if (q.first == "layout") {
                        std::vector<std::string> names; bool present = false;
                        if (q.second.kind == YNode::List) strlist(&q.second, present, names); else names.push_back(q.second.scalar);
                        for (auto &nm : names) {
                            if (nm == "fixed_cuts") c.device.layout_flags |= GE_LAYOUT_FIXED_CUTS;
                            else if (nm == "plain_long_rows") c.device.layout_flags |= GE_LAYOUT_PLAIN_LONG_ROWS;
                            else if (nm == "separate_tables") c.device.layout_flags |= GE_LAYOUT_SEPARATE_TABLES;
                            else if (nm == "packed_records") c.device.layout_flags |= GE_LAYOUT_PACKED_RECORDS;
                            else if (nm == "first_placement") c.device.layout_flags |= GE_LAYOUT_FIRST_PLACEMENT;
                            else if (!nm.empty() && nm != "default") throw std::invalid_argument("device.layout: unknown flag " + nm);
                        }
                    }
                }
            } else c.ignored_keys.push_back(k);
        }
        return c;
    }

    static void check(const Configuration &c) {      // Configuration.check, same order and messages
        const bool hasBca = c.bca.present && c.bca.alpha > 0 && c.bca.epsilon > 0;
        const bool hasOut = c.output.present && (c.output.has_predicate || c.output.has_blank || c.output.has_uri || c.output.has_literal);
        if (!(c.dim > 0)) throw InvalidConfigurationException("No dimension specified");
        if (c.graph.empty()) throw InvalidConfigurationException("No input graph specified");
        if (c.method.empty()) throw InvalidConfigurationException("Invalid method, choose one of: glove, pglove");
        if (!hasBca) throw InvalidConfigurationException("Invalid BCA parameters, alpha and epsilon are mandatory");
        if (!hasOut) throw InvalidConfigurationException("Invalid output parameters, specify at least one group");
    }

    static std::string similarity_to_string(const std::map<std::string, std::string> &m) {     // SimilarityGroup.toString
        auto g = [&](const char *k, const char *def = "") { auto it = m.find(k); return it == m.end() ? std::string(def) : it->second; };
        std::string method = g("method");
        std::string up = method; for (auto &ch : up) ch = (char)std::toupper((unsigned char)ch);
        std::string out = g("sourcePredicate") + " -> " + g("targetPredicate") + "\n method:" + method + ", threshold: " +
                          java_number(std::strtod(g("threshold", "0").c_str(), nullptr), false);
        const std::string smooth = java_number(m.count("smooth") && std::strtod(g("smooth").c_str(), nullptr) != 0 ? std::strtod(g("smooth").c_str(), nullptr) : 1.0, false);
        if (up == "NGRAM_COSINE" || up == "NGRAM_JACCARD") return out + ", ngram: " + (m.count("ngram") && std::atoi(g("ngram").c_str()) ? g("ngram") : std::string("3"));
        if (up == "NUMERIC") return out + ", smooth: " + smooth;
        if (up == "DATE_DAYS" || up == "DATE_MONTHS" || up == "DATE_YEARS")
            return out + ", pattern:" + g("pattern", "iso") + ", smooth: " + smooth + ", time: " + g("time", "bidirectional");
        return out;
    }

    // the settings banner: Main.runProgram's log lines == EmbeddingTextWriter.writeConfig's header (with `prefix`)
    std::vector<std::string> banner() const {
        std::vector<std::string> L;
        L.push_back("Starting the embedding creation process with following settings:");
        L.push_back("Graph File: " + graph);
        L.push_back("Embedding dimensions: " + std::to_string(dim));
        L.push_back("Threads: " + std::to_string(getThreads()));
        L.push_back("BCA Alpha: " + java_number(bca.alpha, false));
        L.push_back("BCA Epsilon: " + java_number(bca.epsilon, false));
        L.push_back(std::string("BCA Directed: ") + (bca.directed ? "true" : "false"));
        L.push_back("BCA normalize: " + getNormalize());
        L.push_back("Gradient Descent Algorithm: " + opt.method);
        L.push_back(method + " Tolerance: " + java_number(opt.tolerance, false));
        L.push_back(method + " Maximum Iterations: " + std::to_string(opt.maxiter));
        if (usingPca()) L.push_back("PCA Minimum Variance: " + java_number(pca.variance, false)); else L.push_back("No PCA will be performed");
        if (usingWeights()) {
            L.push_back("Using weights, predicates that are not listed are ignored:");
            for (auto &w : weights) L.push_back(w.first + ": " + java_number(w.second, true));
        } else L.push_back("No weights specified, using linear weight");
        if (usingSimilarity()) {
            L.push_back("Using the following similarity metrics:");
            for (auto &s : similarity) L.push_back(similarity_to_string(s));
        } else L.push_back("No similarity matching will be performed");
        return L;
    }
};

// Main.createFileName (J/Main.java:80-105)
inline std::string createFileName(const Configuration &c) {
    std::string name = c.graph;
    const size_t slash = name.find_last_of("/\\");
    if (slash != std::string::npos) name = name.substr(slash + 1);
    for (auto &ch : name) ch = (char)std::tolower((unsigned char)ch);
    const size_t dot = name.rfind('.');
    if (dot != std::string::npos) name = name.substr(0, dot);
    std::string m = c.method; for (auto &ch : m) ch = (char)std::tolower((unsigned char)ch);
    name += "_" + m;
    name += c.usingSimilarity() ? "_partial" : "_exact";
    name += c.bca.directed ? "_directed" : "_undirected";
    name += "_" + java_number(c.bca.alpha, false) + "_" + java_number(c.bca.epsilon, false);
    name += "_" + c.opt.method;
    name += (c.usingPca() ? "_pca_" : "_") + std::to_string(c.dim);
    return name;
}

// ------------------------------------------------------------------------------------------------
// Graph ingest: N-Triples subset with Rdf2GrphConverter's rules (the reference reads through Jena;
// Jena is out of scope).  Vertex ids = order of first appearance in the file.
// ------------------------------------------------------------------------------------------------
enum NodeInfo : int8_t { URI = 0, BLANK = 1, LITERAL = 2 };     // J/convert/util/NodeInfo.java

struct InMemoryGraph {
    int32_t V = 0;
    std::vector<std::string> keys;       // vertex label property (dictionary)
    std::vector<int8_t> types;           // vertex type property
    std::vector<int64_t> out_ptr, in_ptr;
    std::vector<int32_t> out_idx, in_idx;
    std::vector<float> out_w, in_w;
    long long triples = 0, skipped = 0;
    long long similarity_pairs = 0;      // "Created links for N literal pairs", summed over the groups
};

// java.lang.String view of a UTF-8 label: UTF-16 code units (malformed bytes become U+FFFD, as Java's decoder does)
inline void utf8_to_utf16(const std::string &in, std::vector<uint16_t> &out) {
    for (size_t i = 0; i < in.size();) {
        const unsigned char c = (unsigned char)in[i];
        uint32_t cp = 0xFFFD; size_t len = 1;
        auto cont = [&](size_t k) { return i + k < in.size() && ((unsigned char)in[i + k] & 0xC0) == 0x80; };
        if (c < 0x80) cp = c;
        else if ((c & 0xE0) == 0xC0 && cont(1)) { cp = ((c & 0x1Fu) << 6) | ((unsigned char)in[i + 1] & 0x3Fu); len = 2; if (cp < 0x80) cp = 0xFFFD; }
        else if ((c & 0xF0) == 0xE0 && cont(1) && cont(2)) { cp = ((c & 0x0Fu) << 12) | (((unsigned char)in[i + 1] & 0x3Fu) << 6) | ((unsigned char)in[i + 2] & 0x3Fu); len = 3; if (cp < 0x800) cp = 0xFFFD; }
        else if ((c & 0xF8) == 0xF0 && cont(1) && cont(2) && cont(3)) {
            cp = ((c & 0x07u) << 18) | (((unsigned char)in[i + 1] & 0x3Fu) << 12) | (((unsigned char)in[i + 2] & 0x3Fu) << 6) | ((unsigned char)in[i + 3] & 0x3Fu); len = 4;
            if (cp < 0x10000 || cp > 0x10FFFF) cp = 0xFFFD;
        }
        if (cp >= 0x10000) { cp -= 0x10000; out.push_back((uint16_t)(0xD800 + (cp >> 10))); out.push_back((uint16_t)(0xDC00 + (cp & 0x3FF))); }
        else out.push_back((uint16_t)cp);
        i += len;
    }
}

// One `similarity:` entry as the converter uses it: CompareGroup (J/compare/CompareGroup.java) + SimilarityGroup.toFunction
struct CompareGroup {
    std::string sourcePredicate, targetPredicate, pattern;
    ge_sim_cfg cfg;
    std::set<int32_t> source, target;        // HashSet<Integer> in the reference; ascending here (only the order of edge insertion depends on it)

    static CompareGroup from(const std::map<std::string, std::string> &m) {
        auto g = [&](const char *k, const char *def = "") { auto it = m.find(k); return it == m.end() ? std::string(def) : it->second; };
        CompareGroup c;
        if (ge_sim_cfg_size() != (int32_t)sizeof(ge_sim_cfg))
            throw std::runtime_error("libgeglove.so was built from another revision of include/geglove.h (ge_sim_cfg differs): rebuild host and library together");
        ge_sim_cfg_default(&c.cfg);
        c.sourcePredicate = g("sourcePredicate"); c.targetPredicate = g("targetPredicate");
        std::string method = g("method"), up = method;
        for (auto &ch : up) ch = (char)std::toupper((unsigned char)ch);
        static const char *const names[] = {"NGRAM_COSINE", "NGRAM_JACCARD", "TOKEN_COSINE", "TOKEN_JACCARD", "JAROWINKLER", "LEVENSHTEIN",
                                            "NUMERIC", "DATE_DAYS", "DATE_MONTHS", "DATE_YEARS"};      // Configuration.java:27-29
        int found = -1;
        for (int k = 0; k < 10; ++k) if (up == names[k]) found = k;
        if (found < 0) throw std::runtime_error("No enum constant org.uu.nl.embedding.util.config.Configuration.SimilarityMethod." + up);   // valueOf
        c.cfg.method = found;
        c.cfg.threshold = std::strtod(g("threshold", "0").c_str(), nullptr);
        c.cfg.ngram = std::atoi(g("ngram", "0").c_str());
        c.cfg.smooth = std::strtod(g("smooth", "0").c_str(), nullptr);
        c.cfg.distance = std::strtod(g("distance", "0").c_str(), nullptr);
        std::string t = g("time", "bidirectional"); for (auto &ch : t) ch = (char)std::toupper((unsigned char)ch);
        if (t == "BACKWARDS") c.cfg.time = GE_TIME_BACKWARDS; else if (t == "FORWARDS") c.cfg.time = GE_TIME_FORWARDS;
        else if (t == "BIDIRECTIONAL") c.cfg.time = GE_TIME_BIDIRECTIONAL;
        else throw std::runtime_error("No enum constant org.uu.nl.embedding.util.config.Configuration.SimilarityGroup.Time." + t);
        c.pattern = g("pattern", "iso");
        if (found >= GE_SIM_DATE_DAYS && !ge_sim_pattern_supported(c.pattern.c_str()))
            throw InvalidConfigurationException("date pattern '" + c.pattern + "' is outside the supported subset (iso, or yyyy/uuuu MM/M dd/d with literals)");
        c.cfg.upper_triangle = c.sourcePredicate == c.targetPredicate;           // Rdf2GrphConverter.java:51
        return c;
    }
};

namespace detail {
struct Term { std::string text; int8_t type; };
inline bool read_term(const std::string &l, size_t &pos, Term &t) {
    while (pos < l.size() && (l[pos] == ' ' || l[pos] == '\t')) ++pos;
    if (pos >= l.size()) return false;
    if (l[pos] == '<') {
        const size_t e = l.find('>', pos);
        if (e == std::string::npos) return false;
        t.text = l.substr(pos + 1, e - pos - 1); t.type = URI; pos = e + 1; return true;
    }
    if (l[pos] == '_' && pos + 1 < l.size() && l[pos + 1] == ':') {
        size_t e = pos; while (e < l.size() && l[e] != ' ' && l[e] != '\t') ++e;
        t.text = l.substr(pos + 2, e - pos - 2); t.type = BLANK; pos = e; return true;
    }
    if (l[pos] == '"') {
        std::string lex; size_t e = pos + 1;
        while (e < l.size() && l[e] != '"') {
            if (l[e] == '\\' && e + 1 < l.size() && (l[e + 1] == 'u' || l[e + 1] == 'U')) {      // \uXXXX / \UXXXXXXXX
                const size_t nd = l[e + 1] == 'u' ? 4 : 8;
                if (e + 2 + nd > l.size()) return false;
                const unsigned long cp = std::strtoul(l.substr(e + 2, nd).c_str(), nullptr, 16);
                if (cp < 0x80) lex.push_back((char)cp);
                else if (cp < 0x800) { lex.push_back((char)(0xC0 | (cp >> 6))); lex.push_back((char)(0x80 | (cp & 0x3F))); }
                else if (cp < 0x10000) { lex.push_back((char)(0xE0 | (cp >> 12))); lex.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); lex.push_back((char)(0x80 | (cp & 0x3F))); }
                else { lex.push_back((char)(0xF0 | (cp >> 18))); lex.push_back((char)(0x80 | ((cp >> 12) & 0x3F))); lex.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); lex.push_back((char)(0x80 | (cp & 0x3F))); }
                e += 2 + nd;
            }
            else if (l[e] == '\\' && e + 1 < l.size()) { const char c = l[e + 1]; lex.push_back(c == 'n' ? '\n' : c == 't' ? '\t' : c == 'r' ? '\r' : c); e += 2; }
            else lex.push_back(l[e++]);
        }
        if (e >= l.size()) return false;
        ++e;
        std::string suffix;
        if (e < l.size() && l[e] == '@') { size_t q = e; while (q < l.size() && l[q] != ' ' && l[q] != '\t') ++q; suffix = l.substr(e, q - e); e = q; }
        else if (e + 1 < l.size() && l[e] == '^' && l[e + 1] == '^') {
            const size_t a = l.find('<', e), b = l.find('>', e);
            if (a == std::string::npos || b == std::string::npos) return false;
            const std::string dt = l.substr(a + 1, b - a - 1);
            if (dt != "http://www.w3.org/2001/XMLSchema#string") suffix = "^^" + dt;      // Node.toString(false) form
            e = b + 1;
        }
        t.text = lex + suffix; t.type = LITERAL; pos = e; return true;
    }
    return false;
}
}  // namespace detail

inline void edges_to_csr(int32_t V, const std::vector<int32_t> &src, const std::vector<int32_t> &dst, const std::vector<float> &w,
                         std::vector<int64_t> &ptr, std::vector<int32_t> &idx, std::vector<float> &wt) {
    // unique neighbours per row, ascending id, first edge in input order wins (getEdge returns the first match)
    std::vector<size_t> order(src.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return src[a] != src[b] ? src[a] < src[b] : dst[a] < dst[b]; });
    ptr.assign((size_t)V + 1, 0); idx.clear(); wt.clear();
    for (size_t q = 0; q < order.size(); ++q) {
        const size_t k = order[q];
        if (q > 0 && src[order[q - 1]] == src[k] && dst[order[q - 1]] == dst[k]) continue;
        idx.push_back(dst[k]); wt.push_back(w[k]); ++ptr[(size_t)src[k] + 1];
    }
    for (int32_t v = 0; v < V; ++v) ptr[(size_t)v + 1] += ptr[(size_t)v];
}

// similarity = true also runs the compare loop of Rdf2GrphConverter.convert (:127-186) on the device (ge_similarity_pairs)
// and adds, per matching pair, the two directed edges weighted with the similarity (:163-173).
inline InMemoryGraph read_ntriples(const std::string &path, const Configuration &cfg, bool similarity = false, int device = 0,
                                   void (*log)(const std::string &) = nullptr) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("Cannot read graph file " + path);
    const std::string ext = path.size() > 3 ? path.substr(path.rfind('.') == std::string::npos ? 0 : path.rfind('.')) : "";
    if (ext == ".tsv" || ext == ".edges") {
        // Plain edge list: one directed edge per line, `source <tab> target [<tab> weight]`, '#' comments.
        // Vertices get consecutive ids in order of first appearance (as Rdf2GrphConverter.addVertex does), all of type URI;
        // there are no predicates, so `weights:` and `similarity:` do not apply.
        InMemoryGraph g;
        std::unordered_map<std::string, int32_t> ids;
        std::vector<int32_t> src, dst; std::vector<float> wt;
        auto vertex = [&](const std::string &name) { auto it = ids.find(name); if (it != ids.end()) return it->second; ids[name] = g.V; g.keys.push_back(name); g.types.push_back(URI); return g.V++; };
        std::string line; long long lineno = 0;
        while (std::getline(f, line)) {
            ++lineno;
            if (!line.empty() && line.back() == '\r') line.pop_back();
            if (line.empty() || line[0] == '#') continue;
            std::vector<std::string> col; size_t a = 0;
            while (a <= line.size()) { const size_t b = line.find('\t', a); col.push_back(line.substr(a, b == std::string::npos ? std::string::npos : b - a)); if (b == std::string::npos) break; a = b + 1; }
            if (col.size() < 2 || col.size() > 3 || col[0].empty() || col[1].empty())
                throw std::runtime_error("graph file " + path + ":" + std::to_string(lineno) + ": expected `source<TAB>target[<TAB>weight]`");
            float w = 1.0f;
            if (col.size() == 3) { char *end = nullptr; w = std::strtof(col[2].c_str(), &end); if (end == col[2].c_str() || *end) throw std::runtime_error("graph file " + path + ":" + std::to_string(lineno) + ": weight is not a number"); }
            ++g.triples;
            const int32_t si = vertex(col[0]), oi = vertex(col[1]);
            src.push_back(si); dst.push_back(oi); wt.push_back(w);
        }
        edges_to_csr(g.V, src, dst, wt, g.out_ptr, g.out_idx, g.out_w);
        edges_to_csr(g.V, dst, src, wt, g.in_ptr, g.in_idx, g.in_w);
        return g;
    }
    if (ext != ".nt" && ext != ".ntriples")
        throw std::runtime_error("graph file " + path + ": only N-Triples (.nt) and tab-separated edge lists (.tsv, .edges) are read natively (the reference parses ttl/trig/hdt through Jena, which is out of scope); convert the file first");
    InMemoryGraph g;
    std::unordered_map<std::string, int32_t> vertexMap;                                      // non-literals
    std::unordered_map<std::string, std::unordered_map<std::string, int32_t>> predLit;       // literals merged PER PREDICATE (:202-213)
    std::unordered_map<std::string, float> weight;
    const bool weighting = cfg.usingWeights();
    for (auto &w : cfg.weights) weight[w.first] = w.second;
    std::vector<int32_t> src, dst; std::vector<float> wt;
    std::vector<CompareGroup> groups;
    std::unordered_map<std::string, size_t> sourceGroups, targetGroups;               // predicate -> group (:41-56; a later entry replaces an earlier one)
    if (similarity) {
        for (auto &m : cfg.similarity) {
            try { groups.push_back(CompareGroup::from(m)); }
            catch (const std::runtime_error &e) {
                // The shipped YAMLs are written in an older dialect (`predicate:` instead of source/targetPredicate, methods
                // `token` / `jaccard`) that this revision's bean cannot load at all.  Such an entry is skipped with a
                // warning, like the other legacy keys; in the current dialect an unknown method fails as valueOf does.
                if (!m.count("predicate") || std::string(e.what()).find("SimilarityMethod") == std::string::npos) throw;
                if (log) log(std::string("skipping legacy similarity entry for ") + m.at("predicate") + ": " + e.what());
            }
        }
        for (size_t k = 0; k < groups.size(); ++k) { sourceGroups[groups[k].sourcePredicate] = k; targetGroups[groups[k].targetPredicate] = k; }
    }
    auto addVertex = [&](const std::string &pred, const detail::Term &n) -> int32_t {     // Rdf2GrphConverter.addVertex
        if (n.type != LITERAL) {
            auto it = vertexMap.find((n.type == BLANK ? "_:" : "") + n.text);
            if (it != vertexMap.end()) return it->second;
        } else {
            auto &m = predLit[pred];
            auto it = m.find(n.text);
            if (it != m.end()) return it->second;
            m[n.text] = g.V;
        }
        if (n.type != LITERAL) vertexMap[(n.type == BLANK ? "_:" : "") + n.text] = g.V;
        g.keys.push_back(n.text); g.types.push_back(n.type);
        return g.V++;
    };
    std::string line;
    while (std::getline(f, line)) {
        size_t pos = 0;
        while (pos < line.size() && (line[pos] == ' ' || line[pos] == '\t')) ++pos;
        if (pos >= line.size() || line[pos] == '#') continue;
        detail::Term s, p, o;
        if (!detail::read_term(line, pos, s) || !detail::read_term(line, pos, p) || !detail::read_term(line, pos, o)) continue;
        ++g.triples;
        float w = 1.0f;
        if (weighting) {                                     // "Ignore unweighted predicates" (:84-90)
            auto it = weight.find(p.text);
            if (it == weight.end()) { ++g.skipped; continue; }
            w = it->second;
        }
        const int32_t si = addVertex(p.text, s), oi = addVertex(p.text, o);
        src.push_back(si); dst.push_back(oi); wt.push_back(w);
        if (!groups.empty()) {                                                       // :100-110, the OBJECT joins the groups of its predicate
            auto a = sourceGroups.find(p.text); if (a != sourceGroups.end()) groups[a->second].source.insert(oi);
            auto b = targetGroups.find(p.text); if (b != targetGroups.end()) groups[b->second].target.insert(oi);
        }
    }
    for (auto &entry : sourceGroups) {                                               // :127-186
        CompareGroup &grp = groups[entry.second];
        if (log) log("Processing similarities for predicate " + entry.first);
        std::vector<int32_t> verts(grp.source.begin(), grp.source.end());
        { std::vector<int32_t> t(grp.target.begin(), grp.target.end()), u; std::set_union(verts.begin(), verts.end(), t.begin(), t.end(), std::back_inserter(u)); verts.swap(u); }
        std::vector<int64_t> offset(1, 0); std::vector<uint16_t> units;
        std::unordered_map<int32_t, int32_t> pos;
        for (int32_t v : verts) { pos[v] = (int32_t)offset.size() - 1; utf8_to_utf16(g.keys[(size_t)v], units); offset.push_back((int64_t)units.size()); }
        if (units.empty()) units.push_back(0);
        std::vector<int32_t> sv(grp.source.begin(), grp.source.end()), tv(grp.target.begin(), grp.target.end()), sp, tp;
        for (int32_t v : sv) sp.push_back(pos[v]);
        for (int32_t v : tv) tp.push_back(pos[v]);
        ge_strings table{(int32_t)verts.size(), offset.data(), units.data()};
        ge_sim_cfg c = grp.cfg;
        c.pattern = grp.pattern == "iso" ? nullptr : grp.pattern.c_str();
        c.device = device;
        ge_sim_pairs *res = nullptr;
        check(ge_similarity_pairs(&table, sp.data(), sv.data(), (int32_t)sp.size(), tp.data(), tv.data(), (int32_t)tp.size(), &c, &res));
        int64_t n = 0; const int32_t *pi = nullptr, *pj = nullptr; const float *ps = nullptr;
        const ge_status st = ge_sim_pairs_get(res, &n, &pi, &pj, &ps);
        if (st != GE_OK) { ge_sim_pairs_destroy(res); check(st); }
        for (int64_t k = 0; k < n; ++k) {
            const int32_t vert = sv[(size_t)pi[k]], other = tv[(size_t)pj[k]];
            src.push_back(vert); dst.push_back(other); wt.push_back(ps[k]);          // e1 = vert -> otherVert
            src.push_back(other); dst.push_back(vert); wt.push_back(ps[k]);          // e2 = otherVert -> vert
        }
        ge_sim_pairs_destroy(res);
        g.similarity_pairs += n;
        if (log) log("Created links for " + std::to_string(n) + " literal pairs");
    }
    edges_to_csr(g.V, src, dst, wt, g.out_ptr, g.out_idx, g.out_w);
    edges_to_csr(g.V, dst, src, wt, g.in_ptr, g.in_idx, g.in_w);
    return g;
}

// ------------------------------------------------------------------------------------------------
// CoOccurrenceMatrix / BookmarkColoring
// ------------------------------------------------------------------------------------------------
struct CoOccurrenceMatrix {                     // J/util/CoOccurrenceMatrix.java:6-17
    virtual ~CoOccurrenceMatrix() = default;
    virtual int vocabSize() const = 0;
    virtual double max() const = 0;
    virtual std::string getKey(int index) const = 0;
    virtual int8_t getType(int index) const = 0;
    virtual int cIdx_I(int i) const = 0;
    virtual int cIdx_J(int j) const = 0;
    virtual float cIdx_C(int i) const = 0;
    virtual int coOccurrenceCount() const = 0;
    virtual void shuffle() = 0;
    // flat views for the native trainer (pre-shuffle order)
    virtual const int32_t *dataI() const = 0;
    virtual const int32_t *dataJ() const = 0;
    virtual const float *dataX() const = 0;
};

class BookmarkColoring : public CoOccurrenceMatrix {        // J/bca/BookmarkColoring.java
public:
    BookmarkColoring(const InMemoryGraph &graph, const Configuration &config, int32_t row_begin = 0, int32_t row_end = 0, int device = -1) : graph_(graph) {
        ge_csr out{graph.V, graph.out_ptr.data(), graph.out_idx.data(), graph.out_w.data()};
        ge_csr in{graph.V, graph.in_ptr.data(), graph.in_idx.data(), graph.in_w.data()};
        ge_bca_cfg cfg{};
        cfg.alpha = config.bca.alpha; cfg.epsilon = config.bca.epsilon; cfg.directed = config.bca.directed ? 1 : 0;
        std::string n = config.getNormalize(); for (auto &ch : n) ch = (char)std::tolower((unsigned char)ch);
        if (n == "none") cfg.normalize = GE_NORM_NONE; else if (n == "unity") cfg.normalize = GE_NORM_UNITY;
        else if (n == "counts") cfg.normalize = GE_NORM_COUNTS; else throw std::invalid_argument("No enum constant BCANormalization." + n);
        cfg.device = device >= 0 ? device : config.device.id;
        cfg.row_begin = row_begin; cfg.row_end = row_end;          // a shard of the bookmarks (multi-GPU); 0,0 = all
        cfg.table_slots = config.device.bca_table_slots; cfg.pool_entries = config.device.bca_pool_entries;
        ge_coo *h = nullptr;
        check(ge_bca_build(&out, &in, &cfg, &h));
        coo_.reset(h);
        check(ge_coo_get(h, &nnz_, &I_, &J_, &X_, nullptr, &max_));
    }
    int vocabSize() const override { return graph_.V; }
    double max() const override { return max_; }
    std::string getKey(int index) const override { return graph_.keys[(size_t)index]; }
    int8_t getType(int index) const override { return graph_.types[(size_t)index]; }
    int cIdx_I(int i) const override { return I_[i]; }
    int cIdx_J(int j) const override { return J_[j]; }
    float cIdx_C(int i) const override { return X_[i]; }
    int coOccurrenceCount() const override { return (int)nnz_; }
    void shuffle() override {}                  // the permutation is part of the native trainer's RNG stream
    const int32_t *dataI() const override { return I_; }
    const int32_t *dataJ() const override { return J_; }
    const float *dataX() const override { return X_; }
private:
    struct Del { void operator()(ge_coo *c) const { ge_coo_destroy(c); } };
    const InMemoryGraph &graph_;
    std::unique_ptr<ge_coo, Del> coo_;
    int64_t nnz_ = 0; const int32_t *I_ = nullptr, *J_ = nullptr; const float *X_ = nullptr; double max_ = 0;
};

// COO checkpoint (the reference has no on-disk form of the matrix, the format is ours):
//   "GECOO1\0\0" | int32 V | int64 nnz | double max | int32 I[nnz] | int32 J[nnz] | float X[nnz] |
//   V x { int8 type, int32 key_len, key bytes }
// It decouples the builder from the trainer: `device.save_coo` writes it after BCA, `device.load_coo` starts from it.
class StoredCooMatrix : public CoOccurrenceMatrix {
public:
    static void save(const std::string &path, const CoOccurrenceMatrix &m) {
        std::ofstream f(path, std::ios::binary);
        if (!f) throw std::runtime_error("cannot write " + path);
        const int32_t V = m.vocabSize(); const int64_t n = m.coOccurrenceCount(); const double mx = m.max();
        f.write("GECOO1\0\0", 8);
        f.write((const char *)&V, 4); f.write((const char *)&n, 8); f.write((const char *)&mx, 8);
        f.write((const char *)m.dataI(), 4 * n); f.write((const char *)m.dataJ(), 4 * n); f.write((const char *)m.dataX(), 4 * n);
        for (int32_t v = 0; v < V; ++v) {
            const int8_t t = m.getType(v); const std::string k = m.getKey(v); const int32_t len = (int32_t)k.size();
            f.write((const char *)&t, 1); f.write((const char *)&len, 4); f.write(k.data(), len);
        }
        if (!f) throw std::runtime_error("short write on " + path);
    }
    explicit StoredCooMatrix(const std::string &path) {
        std::ifstream f(path, std::ios::binary);
        char magic[8];
        if (!f || !f.read(magic, 8) || std::memcmp(magic, "GECOO1\0\0", 8) != 0) throw std::runtime_error(path + " is not a geglove COO checkpoint");
        int64_t n = 0;
        f.read((char *)&V_, 4); f.read((char *)&n, 8); f.read((char *)&max_, 8);
        if (!f || V_ <= 0 || n < 0) throw std::runtime_error("corrupt COO checkpoint " + path);
        I_.resize((size_t)n); J_.resize((size_t)n); X_.resize((size_t)n);
        f.read((char *)I_.data(), 4 * n); f.read((char *)J_.data(), 4 * n); f.read((char *)X_.data(), 4 * n);
        keys_.resize((size_t)V_); types_.resize((size_t)V_);
        for (int32_t v = 0; v < V_; ++v) {
            int32_t len = 0;
            f.read((char *)&types_[(size_t)v], 1); f.read((char *)&len, 4);
            if (!f || len < 0 || len > (1 << 24)) throw std::runtime_error("corrupt COO checkpoint " + path);
            keys_[(size_t)v].resize((size_t)len); f.read(&keys_[(size_t)v][0], len);
        }
        if (!f) throw std::runtime_error("truncated COO checkpoint " + path);
    }
    int vocabSize() const override { return V_; }
    double max() const override { return max_; }
    std::string getKey(int index) const override { return keys_[(size_t)index]; }
    int8_t getType(int index) const override { return types_[(size_t)index]; }
    int cIdx_I(int i) const override { return I_[(size_t)i]; }
    int cIdx_J(int j) const override { return J_[(size_t)j]; }
    float cIdx_C(int i) const override { return X_[(size_t)i]; }
    int coOccurrenceCount() const override { return (int)I_.size(); }
    void shuffle() override {}
    const int32_t *dataI() const override { return I_.data(); }
    const int32_t *dataJ() const override { return J_.data(); }
    const float *dataX() const override { return X_.data(); }
private:
    int32_t V_ = 0; double max_ = 0;
    std::vector<int32_t> I_, J_; std::vector<float> X_;
    std::vector<std::string> keys_; std::vector<int8_t> types_;
};

// ------------------------------------------------------------------------------------------------
// Multi-GPU: `device: {gpus: N}`.  One host thread per rank; rank g uses HIP device (id + g) mod #devices,
// owns a contiguous block of bookmarks / focus rows, and the ranks' context sides are reconciled through ge_sync.
// With one device per rank the exchange runs over RCCL; with fewer devices (a rehearsal) the ranks meet in host memory
// (ge_local_group).  `device.transport: rccl | host` forces either.
// ------------------------------------------------------------------------------------------------
inline void shard_rows(int32_t V, int world, int rank, int32_t &begin, int32_t &end) {
    const int32_t base = V / world, rem = V % world;
    begin = rank * base + std::min(rank, (int)rem);
    end = begin + base + (rank < rem ? 1 : 0);
}
inline int rank_device(const Configuration &config, int rank) {
    const int n = std::max(1, (int)ge_device_count());
    return (config.device.id + rank) % n;
}
template <typename F> void run_ranks(int world, F body) {               // body(rank); the first exception is rethrown
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> err((size_t)world);
    for (int r = 0; r < world; ++r) th.emplace_back([&, r] { try { body(r); } catch (...) { err[(size_t)r] = std::current_exception(); } });
    for (auto &t : th) t.join();
    for (auto &e : err) if (e) std::rethrow_exception(e);
}

// BookmarkColoring built in `gpus` shards of bookmarks, one per rank (no collective: the rows are independent jobs,
// J/bca/BookmarkColoring.java:58-71; max = the max over the shards, :97), concatenated in rank order = bookmark order.
class ShardedBookmarkColoring : public CoOccurrenceMatrix {
public:
    ShardedBookmarkColoring(const InMemoryGraph &graph, const Configuration &config) : graph_(graph) {
        const int world = config.device.gpus;
        std::vector<std::unique_ptr<BookmarkColoring>> part((size_t)world);
        run_ranks(world, [&](int r) {
            int32_t b, e; shard_rows(graph.V, world, r, b, e);
            if (b < e) part[(size_t)r].reset(new BookmarkColoring(graph, config, b, e, rank_device(config, r)));
        });
        for (auto &p : part) if (p) {
            const size_t n = (size_t)p->coOccurrenceCount();
            I_.insert(I_.end(), p->dataI(), p->dataI() + n); J_.insert(J_.end(), p->dataJ(), p->dataJ() + n); X_.insert(X_.end(), p->dataX(), p->dataX() + n);
            max_ = std::max(max_, p->max());
        }
    }
    int vocabSize() const override { return graph_.V; }
    double max() const override { return max_; }
    std::string getKey(int index) const override { return graph_.keys[(size_t)index]; }
    int8_t getType(int index) const override { return graph_.types[(size_t)index]; }
    int cIdx_I(int i) const override { return I_[(size_t)i]; }
    int cIdx_J(int j) const override { return J_[(size_t)j]; }
    float cIdx_C(int i) const override { return X_[(size_t)i]; }
    int coOccurrenceCount() const override { return (int)I_.size(); }
    void shuffle() override {}
    const int32_t *dataI() const override { return I_.data(); }
    const int32_t *dataJ() const override { return J_.data(); }
    const float *dataX() const override { return X_.data(); }
private:
    const InMemoryGraph &graph_;
    std::vector<int32_t> I_, J_; std::vector<float> X_; double max_ = 0;
};

// ------------------------------------------------------------------------------------------------
// Optimizer
// ------------------------------------------------------------------------------------------------
struct Optimum {                                 // J/opt/Optimum.java
    double finalCost = 0;
    std::vector<double> result;
    std::vector<double> costHistory;
};
struct IOptimizer {                              // J/opt/IOptimizer.java:6-11
    virtual ~IOptimizer() = default;
    virtual Optimum optimize() = 0;
    virtual std::string getName() const = 0;
    virtual std::vector<double> extractResult() = 0;
};
enum class CostFunction { GLOVE, PGLOVE };       // GloveCost / PGloveCost

class Adagrad : public IOptimizer {              // J/opt/grad/Adagrad.java + J/opt/Optimizer.java
public:
    Adagrad(const CoOccurrenceMatrix &m, const Configuration &config, CostFunction cf, void (*progress)(int, double, double) = nullptr,
            int opt = GE_OPT_ADAGRAD, const char *name = "Adagrad")
        : coCount_(m.coOccurrenceCount()), vocab_(m.vocabSize()), dim_(config.dim), maxIter_(config.opt.maxiter),
          tolerance_(config.opt.tolerance), progress_(progress), name_(name) {
        if (ge_glove_cfg_size() != (int32_t)sizeof(ge_glove_cfg))
            throw std::runtime_error("libgeglove.so and this host were built from different revisions of include/geglove.h; rebuild both");
        ge_glove_cfg cfg;
        ge_glove_cfg_default(&cfg);
        cfg.opt = opt;
        cfg.vocab_size = vocab_; cfg.dim = dim_; cfg.nnz = coCount_;
        cfg.cost = cf == CostFunction::GLOVE ? GE_COST_GLOVE : GE_COST_PGLOVE;
        cfg.xmax = m.max();
        cfg.threads = config.getThreads();
        // Main seeds from the wall clock (Configuration.setThreadLocalRandom(), J/Main.java:62); `device.seed` pins it
        cfg.seed = config.device.has_seed ? config.device.seed : (long long)(std::time(nullptr)) * 1000LL;
        cfg.mode = config.device.mode == "deterministic" ? GE_MODE_DETERMINISTIC : GE_MODE_HOGWILD;
        cfg.shuffle = config.device.shuffle == "java" ? GE_SHUFFLE_JAVA : config.device.shuffle == "none" ? GE_SHUFFLE_NONE : GE_SHUFFLE_DEVICE;
        cfg.hot_columns = config.device.hot == "none" ? GE_HOT_NONE : config.device.hot == "all" ? GE_HOT_ALL : GE_HOT_AUTO;
        cfg.workers = config.device.workers;
        cfg.emb_dtype = config.device.dtype == "bf16" ? GE_DTYPE_BF16 : GE_DTYPE_F32;
        cfg.device = config.device.id;
        cfg.hot_theta = (float)config.device.hot_theta; cfg.stale_budget = (float)config.device.stale_budget;
        cfg.flush_every = config.device.flush_every; cfg.blocks_per_cu = config.device.blocks_per_cu;
        cfg.layout_flags = config.device.layout_flags;
        ge_glove *h = nullptr;
        check(ge_glove_create(&cfg, m.dataI(), m.dataJ(), m.dataX(), &h));
        h_.reset(h);
    }
    std::string getName() const override { return name_; }
    Optimum optimize() override {                 // Optimizer.optimize, J/opt/Optimizer.java:66-120
        Optimum opt;
        double finalCost = 0, prevCost = 0;
        for (int iteration = 0; iteration < maxIter_; ++iteration) {
            double localCost = 0;
            check(ge_glove_epoch(h_.get(), iteration, &localCost));
            localCost = localCost / coCount_;
            opt.costHistory.push_back(localCost);
            const double iterDiff = std::fabs(prevCost - localCost);
            if (progress_) progress_(iteration, localCost, iterDiff);
            prevCost = localCost;
            if (iterDiff <= tolerance_) { finalCost = localCost; break; }
        }
        opt.result = extractResult();
        opt.finalCost = finalCost;
        return opt;
    }
    std::vector<double> extractResult() override {
        std::vector<double> out((size_t)vocab_ * (size_t)dim_);
        check(ge_glove_extract_f64(h_.get(), out.data()));
        return out;
    }
private:
    struct Del { void operator()(ge_glove *g) const { ge_glove_destroy(g); } };
    std::unique_ptr<ge_glove, Del> h_;
    int coCount_, vocab_, dim_, maxIter_;
    double tolerance_;
    void (*progress_)(int, double, double);
    std::string name_;
};
// `new Adam(...)` / `new AMSGrad(...)` (J/opt/grad/Adam.java, AMSGrad.java): same driver, other update rule in the library
struct Adam : Adagrad {
    Adam(const CoOccurrenceMatrix &m, const Configuration &c, CostFunction cf, void (*pr)(int, double, double) = nullptr) : Adagrad(m, c, cf, pr, GE_OPT_ADAM, "Adam") {}
};
struct AMSGrad : Adagrad {
    AMSGrad(const CoOccurrenceMatrix &m, const Configuration &c, CostFunction cf, void (*pr)(int, double, double) = nullptr) : Adagrad(m, c, cf, pr, GE_OPT_AMSGRAD, "AMSGrad") {}
};

// The trainer over `gpus` ranks (AdaGrad; the exchange's merge rule is defined for it): every rank thread owns a ge_glove
// handle for its block of focus rows and its nonzeros plus a ge_sync; the epoch loop of Optimizer.optimize runs in lockstep,
// the cost of an epoch is the sum over the ranks (Optimizer.java:89-96), and the result is assembled from every rank's focus
// rows and rank 0's context rows after ge_sync_replicate.
class ShardedAdagrad : public IOptimizer {
public:
    ShardedAdagrad(const CoOccurrenceMatrix &m, const Configuration &config, CostFunction cf, void (*progress)(int, double, double) = nullptr)
        : m_(m), config_(config), cf_(cf), progress_(progress) {
        std::string om = config.opt.method; for (auto &ch : om) ch = (char)std::toupper((unsigned char)ch);
        if (om != "ADAGRAD") throw std::invalid_argument("device.gpus > 1 runs opt.method adagrad only (the exchange's merge rule is defined for it)");
        if (config.device.mode == "deterministic") throw std::invalid_argument("device.gpus > 1 needs device.mode hogwild");
    }
    std::string getName() const override { return "Adagrad"; }
    std::vector<double> extractResult() override { return result_; }
    Optimum optimize() override {
        const int world = config_.device.gpus, V = m_.vocabSize(), D = config_.dim;
        const int64_t N = m_.coOccurrenceCount();
        const bool overlap = config_.device.exchange != "sync";
        const bool use_rccl = config_.device.transport == "rccl" || (config_.device.transport == "auto" && ge_device_count() >= world);
        ge_local_group *grp = nullptr;
        unsigned char id[128] = {0};
        if (use_rccl) check(ge_rccl_unique_id(id)); else check(ge_local_group_create(world, &grp));
        struct GroupDel { ge_local_group *g; ~GroupDel() { if (g) ge_local_group_destroy(g); } } group_guard{grp};
        const long long seed = config_.device.has_seed ? config_.device.seed : (long long)(std::time(nullptr)) * 1000LL;
        Optimum opt;
        result_.assign((size_t)V * (size_t)D, 0.0);
        std::vector<double> cost((size_t)world, 0.0);
        std::vector<float> context0;
        Barrier bar(world);
        bool stop = false; double finalCost = 0, prevCost = 0;
        run_ranks(world, [&](int r) {
            try {
            int32_t rb, re; shard_rows(V, world, r, rb, re);
            // the matrix is grouped by row (BookmarkColoring emits bookmark after bookmark): a rank's nonzeros are one range
            const int32_t *I = m_.dataI();
            const int64_t lo = std::lower_bound(I, I + N, rb) - I, hi = std::lower_bound(I, I + N, re) - I;
            if (!std::is_sorted(I, I + N)) throw std::invalid_argument("device.gpus > 1 needs the co-occurrence matrix grouped by row");
            ge_glove_cfg cfg;
            ge_glove_cfg_default(&cfg);
            cfg.vocab_size = V; cfg.dim = D; cfg.nnz = hi - lo;
            cfg.cost = cf_ == CostFunction::GLOVE ? GE_COST_GLOVE : GE_COST_PGLOVE;
            cfg.xmax = m_.max(); cfg.seed = seed; cfg.threads = 1;
            cfg.mode = GE_MODE_HOGWILD; cfg.shuffle = GE_SHUFFLE_DEVICE;
            cfg.hot_columns = config_.device.hot == "none" ? GE_HOT_NONE : config_.device.hot == "all" ? GE_HOT_ALL : GE_HOT_AUTO;
            cfg.workers = config_.device.workers ? config_.device.workers : (overlap && use_rccl ? -256 : 0);   // room for RCCL's kernels beside the epoch
            cfg.emb_dtype = config_.device.dtype == "bf16" ? GE_DTYPE_BF16 : GE_DTYPE_F32;
            cfg.device = rank_device(config_, r);
            cfg.row_begin = rb; cfg.row_end = re;
            cfg.hot_theta = (float)config_.device.hot_theta; cfg.stale_budget = (float)config_.device.stale_budget;
            cfg.flush_every = config_.device.flush_every; cfg.blocks_per_cu = config_.device.blocks_per_cu; cfg.layout_flags = config_.device.layout_flags;
            ge_glove *h = nullptr;
            if (const char *f = std::getenv("GE_HOST_FAIL_RANK")) if (std::atoi(f) == r) throw std::runtime_error("GE_HOST_FAIL_RANK: this rank was told to fail (test hook)");
            check(ge_glove_create(&cfg, m_.dataI() + lo, m_.dataJ() + lo, m_.dataX() + lo, &h));
            std::unique_ptr<ge_glove, void (*)(ge_glove *)> hg(h, ge_glove_destroy);
            bar.wait();                                         // every rank has its handle: only then the collective ge_sync_create (a rank
                                                                // that failed above has abandoned the barrier, and everybody leaves here)
            ge_sync_cfg sc{};
            sc.world = world; sc.rank = r; sc.wire = config_.device.wire == "f32" ? GE_DTYPE_F32 : GE_DTYPE_BF16;
            sc.accum_every = config_.device.accum_every; sc.rccl_id = use_rccl ? id : nullptr; sc.local_group = grp;
            ge_sync *sy = nullptr;
            check(ge_sync_create(h, &sc, &sy));
            std::unique_ptr<ge_sync, void (*)(ge_sync *)> sg(sy, ge_sync_destroy);
            for (int iteration = 0; iteration < config_.opt.maxiter; ++iteration) {
                double c = 0;
                check(ge_sync_epoch(sy, iteration, config_.device.hub_segments, &c));     // the epoch, hub rows reconciled on the way
                // the first two epochs move the model most: exchanged synchronously even in the overlapped form (no bump from landing
                // everything one step late while the busiest rows still grow), the overlap takes over from the third
                check(overlap && iteration >= 2 ? ge_sync_turn(sy) : ge_sync_sync(sy));
                cost[(size_t)r] = c;
                bar.wait();
                if (r == 0) {                                   // Optimizer.optimize's tolerance logic, once for all ranks
                    double localCost = 0; for (double x : cost) localCost += x;
                    localCost /= (double)N;
                    opt.costHistory.push_back(localCost);
                    const double iterDiff = std::fabs(prevCost - localCost);
                    if (progress_) progress_(iteration, localCost, iterDiff);
                    prevCost = localCost;
                    if (iterDiff <= config_.opt.tolerance) { finalCost = localCost; stop = true; }
                }
                bar.wait();
                if (stop) break;
            }
            check(ge_sync_replicate(sy, 0));
            std::vector<float> focus((size_t)(re - rb) * (size_t)D);
            if (re > rb) check(ge_glove_get_state(h, GE_STATE_FOCUS, focus.data(), (int64_t)focus.size()));
            if (r == 0) { context0.resize((size_t)V * (size_t)D); check(ge_glove_get_state(h, GE_STATE_CONTEXT, context0.data(), (int64_t)context0.size())); }
            bar.wait();
            for (size_t k = 0; k < focus.size(); ++k) {         // Optimizer.extractResult: (focus + context) / 2 in fp32, widened
                const size_t g = (size_t)rb * (size_t)D + k;
                result_[g] = (double)((focus[k] + context0[g]) / 2.0f);
            }
            } catch (const std::exception &e) {
                // the peers may be waiting for this rank at the host barrier, in an exchange of the local group, or inside RCCL
                bar.abandon();
                if (grp) ge_local_group_abort(grp);
                std::fprintf(stderr, "rank %d failed: %s\n", r, e.what());
                std::fflush(stderr);
                if (use_rccl) std::_Exit(1);                    // nothing releases a rank blocked in a collective of a communicator one
                throw;                                          // member has left: end the process (Main.java:150-153)
            }
        });
        opt.result = result_;
        opt.finalCost = finalCost;
        return opt;
    }
private:
    struct Barrier {                                            // reusable; abandon() releases everybody when a rank fails
        explicit Barrier(int n) : n_(n) {}
        void wait() {
            std::unique_lock<std::mutex> lk(m_);
            if (dead_) throw std::runtime_error("another rank failed");
            const unsigned long g = gen_;
            if (++count_ == n_) { count_ = 0; ++gen_; cv_.notify_all(); }
            else cv_.wait(lk, [&] { return gen_ != g || dead_; });
            if (dead_) throw std::runtime_error("another rank failed");
        }
        void abandon() { std::lock_guard<std::mutex> lk(m_); dead_ = true; cv_.notify_all(); }
        std::mutex m_; std::condition_variable cv_; int n_, count_ = 0; unsigned long gen_ = 0; bool dead_ = false;
    };
    const CoOccurrenceMatrix &m_;
    const Configuration &config_;
    CostFunction cf_;
    void (*progress_)(int, double, double);
    std::vector<double> result_;
};

// Main.createOptimizer (J/Main.java:107-131)
inline std::unique_ptr<IOptimizer> createOptimizer(const Configuration &config, const CoOccurrenceMatrix &m,
                                                   void (*progress)(int, double, double) = nullptr) {
    std::string method = config.method; for (auto &ch : method) ch = (char)std::toupper((unsigned char)ch);
    CostFunction cf;
    if (method == "GLOVE") cf = CostFunction::GLOVE; else if (method == "PGLOVE") cf = CostFunction::PGLOVE;
    else throw std::invalid_argument("Invalid cost function");
    std::string om = config.opt.method; for (auto &ch : om) ch = (char)std::toupper((unsigned char)ch);
    if (config.device.gpus > 1) return std::unique_ptr<IOptimizer>(new ShardedAdagrad(m, config, cf, progress));
    if (om == "ADAGRAD") return std::unique_ptr<IOptimizer>(new Adagrad(m, config, cf, progress));
    if (om == "ADAM") return std::unique_ptr<IOptimizer>(new Adam(m, config, cf, progress));
    if (om == "AMSGRAD") return std::unique_ptr<IOptimizer>(new AMSGrad(m, config, cf, progress));
    throw std::invalid_argument("Invalid optimization method");
}

// ------------------------------------------------------------------------------------------------
// EmbeddingTextWriter (J/util/write/EmbeddingTextWriter.java)
// ------------------------------------------------------------------------------------------------
class EmbeddingTextWriter {
public:
    EmbeddingTextWriter(const std::string &fileName, const Configuration &config)
        : vectors_(fileName + ".vectors.tsv"), dict_(fileName + ".dict.tsv"), config_(config) {
        write_[URI] = config.output.has_uri; write_[BLANK] = config.output.has_blank; write_[LITERAL] = config.output.has_literal;
    }
    // returns the number of vectors written
    long long write(const Optimum &optimum, const CoOccurrenceMatrix &m, const std::string &outputFolder) {
        std::filesystem::create_directories(outputFolder);      // Files.createDirectories(outputFolder)
        std::ofstream dict(outputFolder + "/" + dict_), vect(outputFolder + "/" + vectors_);
        if (!dict || !vect) throw std::runtime_error("cannot open output files in " + outputFolder);
        for (auto &l : config_.banner()) { dict << "# " << l << "\n"; vect << "# " << l << "\n"; }
        dict << "key\ttype\n";
        const int V = m.vocabSize(), D = config_.dim;
        static const char *names[3] = {"URI", "BLANK", "LITERAL"};
        long long written = 0;
        // which vertices are written (type switch, then the prefix filter of that type: EmbeddingTextWriter.java:100-131), in order
        std::vector<int> keep;
        for (int i = 0; i < V; ++i) {
            const int8_t type = m.getType(i);
            if (!write_[type]) continue;
            const std::vector<std::string> &pre = type == URI ? config_.output.uri : type == BLANK ? config_.output.blank : config_.output.literal;
            if (!pre.empty()) {
                const std::string key = m.getKey(i);
                bool any = false;
                for (auto &p : pre) if (key.compare(0, p.size(), p) == 0) { any = true; break; }
                if (!any) continue;
            }
            keep.push_back(i);
        }
        // the text of the vectors file is 13 bytes per value (60 M values for 300 k vertices at dim 200): blocks of rows are
        // formatted by all host threads, written in order
        const int T = std::max(1, std::min(config_.getThreads() > 1 ? config_.getThreads() : (int)std::thread::hardware_concurrency(), 32));
        const size_t block = 2048;
        std::vector<std::string> part((size_t)T);
        for (size_t base = 0; base < keep.size(); base += block * (size_t)T) {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t) {
                const size_t lo = std::min(keep.size(), base + block * (size_t)t), hi = std::min(keep.size(), lo + block);
                part[(size_t)t].clear();
                if (lo >= hi) continue;
                th.emplace_back([&, t, lo, hi] {
                    std::string &buf = part[(size_t)t];
                    buf.resize((hi - lo) * ((size_t)D * 33 + 1));        // one field is at most 26 characters + a tab
                    size_t w = 0;
                    for (size_t k = lo; k < hi; ++k) {
                        const size_t i = (size_t)keep[k];
                        for (int d = 0; d < D; ++d) { if (d) buf[w++] = '\t'; w += (size_t)java_format_11_6E_to(optimum.result[(size_t)d + i * (size_t)D], &buf[w]); }
                        buf[w++] = '\n';
                    }
                    buf.resize(w);
                });
            }
            for (auto &x : th) x.join();
            for (int t = 0; t < T; ++t) vect.write(part[(size_t)t].data(), (std::streamsize)part[(size_t)t].size());
        }
        for (int i : keep) {
            const std::string key = m.getKey(i);
            std::string clean;
            for (char ch : key) if (ch != '\n' && ch != '\r' && ch != '\t') clean.push_back(ch);
            dict << clean << "\t" << names[m.getType(i)] << "\n";
            ++written;
        }
        return written;
    }
    const std::string &vectorsFile() const { return vectors_; }
    const std::string &dictFile() const { return dict_; }
private:
    std::string vectors_, dict_;
    const Configuration &config_;
    bool write_[3];
};

}  // namespace ge_host
