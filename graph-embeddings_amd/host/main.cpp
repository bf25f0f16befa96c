// geglove -c <config.yml> : the reference's CLI (J/Main.java:133-160) over the MI355X library.
// Same single flag, same settings banner, same output files ./out/<name>.{vectors,dict}.tsv.
#include <ctime>
#include "ge_host.hpp"

using namespace ge_host;

static void log_info(const std::string &who, const std::string &msg) {      // log4j pattern %d{HH:mm:ss} %-5p %-25c{1} :: %m%n
    char ts[16];
    const std::time_t t = std::time(nullptr);
    std::strftime(ts, sizeof ts, "%H:%M:%S", std::localtime(&t));
    std::printf("%s %-5s %-25s :: %s\n", ts, "INFO", who.c_str(), msg.c_str());
    std::fflush(stdout);
}
static void log_error(const std::string &msg) {
    char ts[16];
    const std::time_t t = std::time(nullptr);
    std::strftime(ts, sizeof ts, "%H:%M:%S", std::localtime(&t));
    std::fprintf(stderr, "%s %-5s %-25s :: %s\n", ts, "ERROR", "Graph Embeddings", msg.c_str());
}
static double g_tolerance = 0;
static void progress(int iteration, double cost, double diff) {
    char b[160];
    std::snprintf(b, sizeof b, "epoch %d  cost %.9g  %.6g/%.6g", iteration + 1, cost, diff, g_tolerance);
    log_info("Optimizer", b);
}

static void runProgram(const Configuration &config) {                       // J/Main.java:29-78
    for (auto &l : config.banner()) log_info("Graph Embeddings", l);
    for (auto &k : config.ignored_keys) log_info("Configuration", "ignoring key not known to this revision's bean: " + k);
    std::string outFileName = config.output.name;
    if (outFileName.empty()) outFileName = createFileName(config);
    log_info("Graph Embeddings", "Writing files with prefix: " + outFileName);

    std::unique_ptr<InMemoryGraph> graph;
    std::unique_ptr<CoOccurrenceMatrix> matrix;
    if (!config.device.load_coo.empty()) {
        matrix.reset(new StoredCooMatrix(config.device.load_coo));
        log_info("BookmarkColoring", "loaded COO checkpoint " + config.device.load_coo);
    } else {
        if (!config.usingSimilarity()) log_info("Rdf2GrphConverter", "Partial matching is disabled, no edges between similar literals are added");
        graph.reset(new InMemoryGraph(read_ntriples(config.graph, config, true, config.device.id,
                                                    [](const std::string &m) { log_info("Rdf2GrphConverter", m); })));
        char b[200];
        std::snprintf(b, sizeof b, "Skipped %lld unweighted triples (%.2f %%)", graph->skipped,
                      graph->triples ? 100.0 * (double)graph->skipped / (double)graph->triples : 0.0);
        log_info("Rdf2GrphConverter", b);
        std::snprintf(b, sizeof b, "graph: %d vertices, %zu out-neighbour pairs", graph->V, graph->out_idx.size());
        log_info("Rdf2GrphConverter", b);
        if (config.device.gpus > 1) matrix.reset(new ShardedBookmarkColoring(*graph, config));
        else matrix.reset(new BookmarkColoring(*graph, config));
        if (!config.device.save_coo.empty()) {
            StoredCooMatrix::save(config.device.save_coo, *matrix);
            log_info("BookmarkColoring", "wrote COO checkpoint " + config.device.save_coo);
        }
    }
    const CoOccurrenceMatrix &bca = *matrix;
    {
        char b[200];
        std::snprintf(b, sizeof b, "BCA: %d co-occurrences, max %.9g", bca.coOccurrenceCount(), bca.max());
        log_info("BookmarkColoring", b);
    }
    g_tolerance = config.opt.tolerance;
    std::unique_ptr<IOptimizer> optimizer = createOptimizer(config, bca, progress);
    const Optimum optimum = optimizer->optimize();
    {
        char b[120];
        std::snprintf(b, sizeof b, "Finished optimization with final cost: %.12g", optimum.finalCost);
        log_info("Optimizer", b);
    }
    EmbeddingTextWriter writer(outFileName, config);
    const long long n = writer.write(optimum, bca, "out");
    log_info("EmbeddingTextWriter", "wrote " + std::to_string(n) + " vectors to out/" + writer.vectorsFile());
}

int main(int argc, char **argv) {
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "-c") {
            if (i + 1 < argc) {
                try {
                    std::ifstream probe(argv[i + 1]);
                    if (!probe) { log_error(std::string("Cannot find configuration file + ") + argv[i + 1]); return 1; }
                    probe.close();
                    const Configuration config = Configuration::load(argv[i + 1]);
                    Configuration::check(config);
                    runProgram(config);
                } catch (const std::exception &e) {
                    log_error(e.what());
                    return 1;
                }
            } else {
                log_error("No configuration file specified, exiting...");
                return 1;
            }
        }
    }
    return 0;
}
