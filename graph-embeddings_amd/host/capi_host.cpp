// Small C entry points over ge_host.hpp so that the Python tests can exercise the C++ host logic
// (YAML subset, bean check, file name, banner, number formatting, N-Triples ingest) without a GPU.
#include "ge_host.hpp"
using namespace ge_host;
static thread_local std::string g_buf;
extern "C" {
const char *geh_format_11_6E(double v) { g_buf = java_format_11_6E(v); return g_buf.c_str(); }
const char *geh_java_double(double v) { g_buf = java_number(v, false); return g_buf.c_str(); }
const char *geh_java_float(float v) { g_buf = java_number(v, true); return g_buf.c_str(); }
// returns "" on success, the exception text otherwise; *out receives banner + file name + ignored keys, one per line
const char *geh_config_summary(const char *path, int do_check) {
    try {
        const Configuration c = Configuration::load(path);
        if (do_check) Configuration::check(c);
        std::string s;
        for (auto &l : c.banner()) s += "# " + l + "\n";
        s += "name=" + createFileName(c) + "\n";
        for (auto &k : c.ignored_keys) s += "ignored=" + k + "\n";
        s += "uri=" + std::to_string(c.output.has_uri) + ":" + std::to_string(c.output.uri.size()) + "\n";
        g_buf = "OK\n" + s;
    } catch (const std::exception &e) { g_buf = std::string("ERR\n") + e.what(); }
    return g_buf.c_str();
}
// parses an N-Triples file with the weights of a config; writes counts and the CSR into caller buffers via a text dump
static const char *graph_summary(const char *config_path, const char *nt_path, bool similarity);
const char *geh_graph_summary(const char *config_path, const char *nt_path) { return graph_summary(config_path, nt_path, false); }
// the same with the similarity edges of the config's `similarity:` groups (needs the GPU: ge_similarity_pairs)
const char *geh_graph_summary_similarity(const char *config_path, const char *nt_path) { return graph_summary(config_path, nt_path, true); }
static const char *graph_summary(const char *config_path, const char *nt_path, bool similarity) {
    try {
        const Configuration c = Configuration::load(config_path);
        const InMemoryGraph g = read_ntriples(nt_path, c, similarity);
        std::string s = "V=" + std::to_string(g.V) + " triples=" + std::to_string(g.triples) + " skipped=" + std::to_string(g.skipped) +
                        (similarity ? " pairs=" + std::to_string(g.similarity_pairs) : std::string()) + "\n";
        for (int v = 0; v < g.V; ++v) {
            s += std::to_string(v) + "\t" + std::to_string((int)g.types[(size_t)v]) + "\t" + g.keys[(size_t)v] + "\tout:";
            for (int64_t k = g.out_ptr[(size_t)v]; k < g.out_ptr[(size_t)v + 1]; ++k) s += " " + std::to_string(g.out_idx[(size_t)k]) + "(" + java_number(g.out_w[(size_t)k], true) + ")";
            s += "\tin:";
            for (int64_t k = g.in_ptr[(size_t)v]; k < g.in_ptr[(size_t)v + 1]; ++k) s += " " + std::to_string(g.in_idx[(size_t)k]);
            s += "\n";
        }
        g_buf = "OK\n" + s;
    } catch (const std::exception &e) { g_buf = std::string("ERR\n") + e.what(); }
    return g_buf.c_str();
}
}
