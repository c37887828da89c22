// cli_main.cpp — the `FDES` command line (src/FDES.cu:61-263, Useage.txt:26-37): same flags,
// same defaults, same side-effect files; errors are reported through the exit status.
#include <getopt.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/fdes_abi.h"

static void usage()
{
    std::fprintf(stderr,
                 "\nUsage:\n"
                 "  FDES [--input_name <file.cnf | file.emd | file.qsc>]   simulation parameters (default dataFDES.cnf)\n"
                 "       [--image_name <file>]    raw float32 images (default Measurements.bin)\n"
                 "       [--emd_name <file>]      EMD/HDF5 results (default results.emd)\n"
                 "       [--print_level <0|1|2>]  0 images, 1 + potential slices, 2 + exit waves\n"
                 "       [--gpu_index <n>]        device to run on (default 0)\n"
                 "       [--help] [--version]\n");
}

int main(int argc, char** argv)
{
    std::string input = "", image = "Measurements.bin", emd = "results.emd";
    int gpu = 0, print_level = 0;
    static struct option opts[] = {{"gpu_index", required_argument, 0, 0},  {"input_name", required_argument, 0, 0},
                                   {"image_name", required_argument, 0, 0}, {"emd_name", required_argument, 0, 0},
                                   {"print_level", required_argument, 0, 0}, {"help", no_argument, 0, 0},
                                   {"version", no_argument, 0, 0},          {NULL, 0, 0, 0}};
    for (;;) {
        int idx = 0;
        int c = getopt_long(argc, argv, "", opts, &idx);
        if (c == -1) break;
        if (c != 0) { usage(); return EXIT_FAILURE; }
        switch (idx) {
        case 0: gpu = std::atoi(optarg); break;
        case 1: input = optarg; break;
        case 2: image = optarg; break;
        case 3: emd = optarg; break;
        case 4: print_level = std::atoi(optarg); break;
        case 5: usage(); return EXIT_FAILURE;
        case 6: std::fprintf(stderr, "\n FDES Version : 0.1 (MI355X engine, ABI %d)\n", fdes_abi_version()); return EXIT_FAILURE;
        }
    }
    if (print_level < 0 || print_level > 2) { std::fprintf(stderr, "\n printLevel error\n"); return EXIT_FAILURE; }
    if (input.empty()) {
        std::fprintf(stderr, "  confOption is not set, using default configuration\n");
        input = "dataFDES.cnf"; // first dataFDES.cnf, then config.emd (src/FDES.cu:167-188)
        if (FILE* t = std::fopen(input.c_str(), "rb")) std::fclose(t);
        else input = "config.emd";
    }
    int rc = fdes_run_file(gpu, print_level, input.c_str(), image.c_str(), emd.c_str(), nullptr, 0, nullptr);
    if (rc) { std::fprintf(stderr, "  FDES failed (%d)\n", rc); return EXIT_FAILURE; }
    std::fprintf(stderr, "  Done.\n");
    return EXIT_SUCCESS;
}
