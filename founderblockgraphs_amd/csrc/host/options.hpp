// options.hpp -- command line of the founderblockgraph host program.
// Contract taken from the reference's gengetopt specification (founderblockgraph_cmdline.ggo:17-37)
// and the behaviour of its generated parser (founderblockgraph_cmdline.c:169-217, 446-462, 510-560):
// same option names, short forms, defaults, help/version text, error messages and exit codes.
#pragma once
#include <string>

struct Options {
    std::string input, output, graphviz_output, memory_chart_output, ignore_chars;
    bool input_given = false, output_given = false, graphviz_output_given = false,
         memory_chart_output_given = false, ignore_chars_given = false;
    long gap_limit = 1;              // default=`1'
    long threads = -1;               // default=`-1'
    long heuristic_subset = -1;      // hidden, default=`-1'
    bool elastic = false, gfa = false, output_paths = false, disable_elastic_tricks = false;
};

// Returns 0 on success, non-zero (EXIT_FAILURE) after printing the parser's own message.
// --help / --full-help / --version print to stdout and exit(EXIT_SUCCESS) like the reference.
int parse_options(int argc, char **argv, Options &opt);
