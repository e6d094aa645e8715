#include "fasta.hpp"
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>

namespace {

// fbg.cpp:136-149
bool length_ok(const std::string &identifier, const std::string &seq, std::size_t expected)
{
    if (seq.size() == expected) return true;
    std::cerr << "WARNING: length of the sequence \xe2\x80\x9c" << (identifier.empty() ? identifier : identifier.substr(1))
              << "\xe2\x80\x9d does not match that of the first sequence; skipping. (" << expected << " vs. "
              << seq.size() << ")\n";
    return false;
}

// fbg.cpp:103-133: only '-' runs are examined (the help text also mentions N's; the code does not)
bool gaps_ok(const std::string &identifier, const std::string &seq, std::size_t gap_limit)
{
    if (gap_limit == 0) return true;
    std::size_t run = 0, longest = 0;
    for (const char c : seq) {
        if (c == '-') ++run;
        else { longest = std::max(run, longest); run = 0; }
    }
    longest = std::max(run, longest);
    if (longest < gap_limit) return true;
    std::cerr << "NOTICE: Sequence \xe2\x80\x9c" << (identifier.empty() ? identifier : identifier.substr(1))
              << "\xe2\x80\x9d contained a gap run with " << longest << " characters.\n";
    return false;
}

} // namespace

bool read_msa(const std::string &path, long gap_limit, bool elastic, bool output_paths, Msa &msa)
{
    std::ifstream fs(path, std::ios::in | std::ios::binary);
    if (!fs) return false;
    std::string line, identifier, entry;
    std::vector<std::string> rows;
    if (!std::getline(fs, identifier)) return false;       // first line is taken as a header (fbg.cpp:160)
    if (output_paths) msa.identifiers.push_back(identifier.empty() ? identifier : identifier.substr(1));
    std::size_t expected = 0;
    bool first = true;
    auto finish_record = [&]() {
        if (first) { expected = entry.size(); first = false; }
        if (length_ok(identifier, entry, expected) &&
            (elastic || gaps_ok(identifier, entry, (std::size_t)gap_limit)))
            rows.push_back(entry);
    };
    while (std::getline(fs, line)) {
        if (!line.empty() && line[0] == '>') {
            if (output_paths) msa.identifiers.push_back(line.substr(1));
            finish_record();
            entry.clear();
            identifier = line;
        } else {
            entry += line;                                   // verbatim: no \r stripping, no case folding
        }
    }
    finish_record();
    msa.m = rows.size();
    msa.n = rows.empty() ? 0 : rows[0].size();
    msa.cells.resize(msa.m * msa.n);
    for (uint64_t i = 0; i < msa.m; i++) std::copy(rows[i].begin(), rows[i].end(), msa.cells.begin() + i * msa.n);
    return true;
}
