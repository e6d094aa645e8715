// xgfa.hpp -- xGFA writer producing the bytes of output_efg (founderblockgraph.cpp:1185-1301,
// SURVEY.md Appendix A.3), and the graph statistics segment() prints (fbg.cpp:667-728).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "fasta.hpp"

// boundaries: inclusive block ends, last one == n (fbg.cpp:2027-2039).  Returns false on I/O failure.
bool write_xgfa(const Msa &msa, const std::vector<uint64_t> &boundaries, bool output_paths,
                const std::string &path, std::string &error);

// The same bytes from the node / edge arrays of fbg_block_graph (include/fbg_hip.h): nothing is hashed here, the
// labels of the S lines are cut out of the representative rows.
struct BlockGraph {
    std::vector<uint32_t> node_of, rep_row;     // [nb * m]
    std::vector<uint64_t> first_node;           // [nb + 1]
    std::vector<uint64_t> edge_count, edges;    // [nb], [nb * m]
};
// --threads=T of the reference sets the threads of its scan (fbg.cpp:3395); the scan runs on the GPU here, so T
// bounds the host threads that format the xGFA instead (0 = machine default).
void set_host_threads(unsigned t);
bool write_xgfa_graph(const Msa &msa, const std::vector<uint64_t> &boundaries, const BlockGraph &g, bool output_paths,
                      const std::string &path, std::string &error);

struct GraphStats { uint64_t nodes = 0, total_label_length = 0, founders = 0, edges = 0; };
GraphStats segment_stats(const Msa &msa, const std::vector<uint64_t> &boundaries);
