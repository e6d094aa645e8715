// options_dump.cpp -- test helper: parses argv with the host program's parser and prints the
// fields in the format of oracle/refcmdline_main.c, so tests can diff the two parsers.
#include <cstdio>
#include <cstdlib>
#include "options.hpp"

int main(int argc, char **argv)
{
    Options o;
    if (parse_options(argc, argv, o) != 0) return EXIT_FAILURE;
    std::printf("input=%s\noutput=%s\ngap_limit=%ld\ngraphviz=%s\nmemchart=%s\nelastic=%d\ngfa=%d\npaths=%d\n"
                "ignore=%s\nthreads=%ld\nheuristic=%ld\nnotricks=%d\n",
                o.input.c_str(), o.output.c_str(), o.gap_limit, o.graphviz_output.c_str(),
                o.memory_chart_output.c_str(), (int)o.elastic, (int)o.gfa, (int)o.output_paths,
                o.ignore_chars.c_str(), o.threads, o.heuristic_subset, (int)o.disable_elastic_tricks);
    return 0;
}
