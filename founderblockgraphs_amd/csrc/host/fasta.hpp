// fasta.hpp -- MSA input with the rules of read_input / check_gaps / check_sequence_length
// (founderblockgraph.cpp:103-201).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

struct Msa {
    uint64_t m = 0, n = 0;
    std::vector<uint8_t> cells;            // row-major m x n
    std::vector<std::string> identifiers;  // header text after '>', one per header line (only with -p)
};

// Returns false when the file cannot be read at all.  Rows are dropped with the reference's
// WARNING / NOTICE lines on stderr; msa.m == 0 afterwards means "Unable to read sequences".
bool read_msa(const std::string &path, long gap_limit, bool elastic, bool output_paths, Msa &msa);
