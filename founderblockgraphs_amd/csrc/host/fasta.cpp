#include "fasta.hpp"
#include <algorithm>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <string_view>

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
    // the whole file in one buffer, lines cut in place (std::getline's rules: split at '\n', a last line without one
    // counts, nothing else is stripped); records of a single line -- the usual case -- are copied once, into the matrix
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    std::string buf;
    {
        long sz = -1;
        if (std::fseek(fp, 0, SEEK_END) == 0) { sz = std::ftell(fp); std::rewind(fp); }
        if (sz > 0) {                                      // a regular file: one read straight into the buffer
            buf.resize((size_t)sz);
            const size_t got = std::fread(&buf[0], 1, (size_t)sz, fp);
            buf.resize(got);
        }
        char tmp[1 << 16];                                 // pipes, or a file that grew meanwhile
        size_t got;
        while ((got = std::fread(tmp, 1, sizeof(tmp), fp)) > 0) buf.append(tmp, got);
        std::fclose(fp);
    }
    msa.cells.clear();
    msa.cells.reserve(buf.size());                         // an upper bound: no reallocation while rows are appended
    size_t pos = 0;
    bool have_line = false;
    auto next_line = [&](std::string_view &line) -> bool {
        if (pos >= buf.size()) return false;
        const size_t nl = buf.find('\n', pos);
        const size_t end = nl == std::string::npos ? buf.size() : nl;
        line = std::string_view(buf.data() + pos, end - pos);
        pos = nl == std::string::npos ? buf.size() : nl + 1;
        return true;
    };
    std::string_view line;
    have_line = next_line(line);
    if (!have_line) return false;                          // first line is taken as a header (fbg.cpp:160)
    std::string identifier(line);
    if (output_paths) msa.identifiers.push_back(identifier.empty() ? identifier : identifier.substr(1));
    std::size_t expected = 0;
    bool first = true;
    uint64_t rows = 0;
    std::string entry;                                     // records spread over several lines are joined here
    std::string_view single;                               // a record of one line lives in the buffer
    int lines_in_entry = 0;
    auto finish_record = [&]() {
        if (lines_in_entry > 1 || lines_in_entry == 0) single = std::string_view(entry);
        if (first) { expected = single.size(); first = false; }
        bool keep = true;
        if (single.size() != expected) keep = length_ok(identifier, std::string(single), expected);
        if (keep && !elastic) keep = gaps_ok(identifier, std::string(single), (std::size_t)gap_limit);
        if (keep) {
            msa.cells.insert(msa.cells.end(), reinterpret_cast<const uint8_t *>(single.data()),
                             reinterpret_cast<const uint8_t *>(single.data()) + single.size());
            rows++;
        }
    };
    while (next_line(line)) {
        if (!line.empty() && line[0] == '>') {
            if (output_paths) msa.identifiers.push_back(std::string(line.substr(1)));
            finish_record();
            entry.clear();
            lines_in_entry = 0;
            identifier.assign(line.data(), line.size());
        } else {
            if (lines_in_entry == 0) single = line;        // verbatim: no \r stripping, no case folding
            else {
                if (lines_in_entry == 1) entry.assign(single.data(), single.size());
                entry.append(line.data(), line.size());
            }
            lines_in_entry++;
        }
    }
    finish_record();
    msa.m = rows;
    msa.n = rows == 0 ? 0 : expected;
    return true;
}
