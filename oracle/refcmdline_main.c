/*
 * refcmdline_main.c -- TEST INFRASTRUCTURE.  Driver around the reference's own generated option
 * parser (founderblockgraph_cmdline.c, compiled from /root/reference where it lies; see Makefile
 * target _ref/refcmdline).  It parses argv exactly as the reference's main() does
 * (founderblockgraph.cpp:3302-3304) and dumps the parsed fields, so that tests can pin the command
 * line contract of the C++ host program against the real reference parser.
 */
#include <stdio.h>
#include <stdlib.h>
#include "founderblockgraph_cmdline.h"

int main(int argc, char **argv)
{
    struct gengetopt_args_info a;
    if (0 != cmdline_parser(argc, argv, &a)) return EXIT_FAILURE;
    printf("input=%s\noutput=%s\ngap_limit=%ld\ngraphviz=%s\nmemchart=%s\nelastic=%d\ngfa=%d\npaths=%d\n"
           "ignore=%s\nthreads=%ld\nheuristic=%ld\nnotricks=%d\n",
           a.input_arg, a.output_arg, a.gap_limit_arg, a.graphviz_output_given ? a.graphviz_output_arg : "",
           a.memory_chart_output_given ? a.memory_chart_output_arg : "", a.elastic_flag, a.gfa_flag,
           a.output_paths_flag, a.ignore_chars_arg ? a.ignore_chars_arg : "", a.threads_arg,
           a.heuristic_subset_arg, a.disable_elastic_tricks_flag);
    return 0;
}
