#include "options.hpp"
#include <getopt.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

const char *const kPackage = "founderblockgraphs";
const char *const kVersion = "0.5";
const char *const kUsage =
    "Usage: founderblockgraph --input=MSA.fasta --output={MSA.index|efg.xgfa} [--gfa]\n"
    "[--elastic] [--gap-limit=GAPLIMIT] [--threads=THREADNUM]\n"
    "[--graphviz-output=efg.dot] [--output-paths] [--ignore-chars=\"ALPHABET\"]";
const char *const kPurpose = "Constructs a semi-repeat-free (Elastic) Founder Graph";
const char *const kDescription =
    "Input is MSA given in fasta format. In standard mode (without --elastic), rows\n"
    "with runs of gaps \xe2\x80\x98-\xe2\x80\x99 or N\xe2\x80\x99s \xe2\x89\xa5 GAPLIMIT will be filtered out.";

struct HelpLine { const char *text; bool hidden; };
const HelpLine kHelp[] = {
    {"  -h, --help                    Print help and exit", false},
    {"      --full-help               Print help, including hidden options, and exit", false},
    {"  -V, --version                 Print version and exit", false},
    {"      --input=filename          MSA input path", false},
    {"      --output=filename         Index/EFG output path", false},
    {"      --gap-limit=GAPLIMIT      Gap limit (suppressed by --elastic)\n"
     "                                  (default=`1')", false},
    {"      --graphviz-output=filename\n"
     "                                Graphviz output path", false},
    {"      --memory-chart-output=filename\n"
     "                                Memory chart output path", false},
    {"  -e, --elastic                 Min-max-length semi-repeat-free segmentation\n"
     "                                  (default=off)", false},
    {"      --gfa                     Saves output in xGFA format  (default=off)", false},
    {"  -p, --output-paths            Print the original sequences as paths of the\n"
     "                                  xGFA graph (requires --gfa)  (default=off)", false},
    {"      --ignore-chars=STRING     Ignore these characters for the indexability\n"
     "                                  property/pattern matching", false},
    {"  -t, --threads=THREADNUM       Max # threads  (default=`-1')", false},
    {"      --heuristic-subset=ROWNUM To save memory, compute the optimal\n"
     "                                  segmentation in chunks of ROWNUM MSA rows,\n"
     "                                  then fix the resulting graph iteratively,\n"
     "                                  sacrificing optimality  (default=`-1')", true},
    {"      --disable-elastic-tricks  Disable the tricks considering the start and\n"
     "                                  end of sequences as unique  (default=off)", true},
};

void print_help(bool full)
{
    std::printf("%s\n%s\n\n%s\n\n", kUsage, kPurpose, kDescription);
    for (const HelpLine &h : kHelp)
        if (full || !h.hidden) std::printf("%s\n", h.text);
}

enum Id { ID_FULL_HELP = 256, ID_INPUT, ID_OUTPUT, ID_GAP_LIMIT, ID_GRAPHVIZ, ID_MEMCHART, ID_GFA,
          ID_IGNORE, ID_HEURISTIC, ID_NO_TRICKS };

bool once(const char *prog, unsigned &seen, const char *long_opt, char short_opt)
{
    if (seen) {
        if (short_opt != '-')
            std::fprintf(stderr, "%s: `--%s' (`-%c') option given more than once\n", prog, long_opt, short_opt);
        else
            std::fprintf(stderr, "%s: `--%s' option given more than once\n", prog, long_opt);
        return false;
    }
    seen++;
    return true;
}

bool to_long(const char *prog, const char *val, long &out)
{
    char *stop = nullptr;
    out = std::strtol(val, &stop, 0);
    if (!(stop && *stop == '\0')) {
        std::fprintf(stderr, "%s: invalid numeric value: %s\n", prog, val);
        return false;
    }
    return true;
}

} // namespace

int parse_options(int argc, char **argv, Options &opt)
{
    static const struct option long_options[] = {
        {"help", 0, nullptr, 'h'},
        {"full-help", 0, nullptr, ID_FULL_HELP},
        {"version", 0, nullptr, 'V'},
        {"input", 1, nullptr, ID_INPUT},
        {"output", 1, nullptr, ID_OUTPUT},
        {"gap-limit", 1, nullptr, ID_GAP_LIMIT},
        {"graphviz-output", 1, nullptr, ID_GRAPHVIZ},
        {"memory-chart-output", 1, nullptr, ID_MEMCHART},
        {"elastic", 0, nullptr, 'e'},
        {"gfa", 0, nullptr, ID_GFA},
        {"output-paths", 0, nullptr, 'p'},
        {"ignore-chars", 1, nullptr, ID_IGNORE},
        {"threads", 1, nullptr, 't'},
        {"heuristic-subset", 1, nullptr, ID_HEURISTIC},
        {"disable-elastic-tricks", 0, nullptr, ID_NO_TRICKS},
        {nullptr, 0, nullptr, 0}};
    const char *prog = argv[0];
    unsigned seen[16] = {0};
    optarg = nullptr; optind = 0; opterr = 1; optopt = '?';
    for (;;) {
        int idx = 0;
        const int c = getopt_long(argc, argv, "hVept:", long_options, &idx);
        if (c == -1) break;
        switch (c) {
        case 'h': print_help(false); std::exit(EXIT_SUCCESS);
        case ID_FULL_HELP: print_help(true); std::exit(EXIT_SUCCESS);
        case 'V': std::printf("%s %s\n", kPackage, kVersion); std::exit(EXIT_SUCCESS);
        case ID_INPUT:
            if (!once(prog, seen[0], "input", '-')) return EXIT_FAILURE;
            opt.input = optarg; opt.input_given = true; break;
        case ID_OUTPUT:
            if (!once(prog, seen[1], "output", '-')) return EXIT_FAILURE;
            opt.output = optarg; opt.output_given = true; break;
        case ID_GAP_LIMIT:
            if (!once(prog, seen[2], "gap-limit", '-') || !to_long(prog, optarg, opt.gap_limit)) return EXIT_FAILURE;
            break;
        case ID_GRAPHVIZ:
            if (!once(prog, seen[3], "graphviz-output", '-')) return EXIT_FAILURE;
            opt.graphviz_output = optarg; opt.graphviz_output_given = true; break;
        case ID_MEMCHART:
            if (!once(prog, seen[4], "memory-chart-output", '-')) return EXIT_FAILURE;
            opt.memory_chart_output = optarg; opt.memory_chart_output_given = true; break;
        case 'e':
            if (!once(prog, seen[5], "elastic", 'e')) return EXIT_FAILURE;
            opt.elastic = !opt.elastic; break;
        case ID_GFA:
            if (!once(prog, seen[6], "gfa", '-')) return EXIT_FAILURE;
            opt.gfa = !opt.gfa; break;
        case 'p':
            if (!once(prog, seen[7], "output-paths", 'p')) return EXIT_FAILURE;
            opt.output_paths = !opt.output_paths; break;
        case ID_IGNORE:
            if (!once(prog, seen[8], "ignore-chars", '-')) return EXIT_FAILURE;
            opt.ignore_chars = optarg; opt.ignore_chars_given = true; break;
        case 't':
            if (!once(prog, seen[9], "threads", 't') || !to_long(prog, optarg, opt.threads)) return EXIT_FAILURE;
            break;
        case ID_HEURISTIC:
            if (!once(prog, seen[10], "heuristic-subset", '-') || !to_long(prog, optarg, opt.heuristic_subset))
                return EXIT_FAILURE;
            break;
        case ID_NO_TRICKS:
            if (!once(prog, seen[11], "disable-elastic-tricks", '-')) return EXIT_FAILURE;
            opt.disable_elastic_tricks = !opt.disable_elastic_tricks; break;
        case '?':   // getopt_long already printed its message
            return EXIT_FAILURE;
        default:
            std::fprintf(stderr, "%s: option unknown: %c\n", kPackage, c);
            std::abort();
        }
    }
    int err = 0;
    if (!opt.input_given) { std::fprintf(stderr, "%s: '--input' option required\n", prog); err = 1; }
    if (!opt.output_given) { std::fprintf(stderr, "%s: '--output' option required\n", prog); err = 1; }
    return err ? EXIT_FAILURE : 0;
}
