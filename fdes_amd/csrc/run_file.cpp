// run_file.cpp — file-level entry points: the legacy `FDES(...)` export of the reference's
// shared library (src/FDESExport.cu:59-178) and its int-returning twin, shared with the CLI
// (src/FDES.cu:61-263).  Host C++ only; the GPU work happens behind fdes_build_measurements.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "fdes_internal.h"

namespace {
bool has_ext(const char* name, const char* ext) { return std::strstr(name, ext) != nullptr; }
void print_progress(void*, int64_t done, int64_t total)
{
    if (total <= 0) return;
    std::fprintf(stderr, "\r Progress: %d%%%s", (int)(100 * done / total), done == total ? "\n" : "");
    std::fflush(stderr);
}
}

extern "C" int fdes_run_file(int gpu_index, int print_level, const char* input_name, const char* image_name,
                             const char* emd_name, const float* atomsArray, int numAtoms, float* dstImage)
{
    if (!input_name) return FDES_EINVAL;
    if (print_level < 0 || print_level > 2) return FDES_EINVAL;
    // dispatch by extension (src/FDES.cu:107-118, src/FDESExport.cu:84-101)
    const bool is_emd = has_ext(input_name, ".emd");
    const bool is_qsc = !is_emd && !has_ext(input_name, ".cnf") && has_ext(input_name, ".qsc");
    if (!is_emd && !is_qsc && !has_ext(input_name, ".cnf")) {
        std::fprintf(stderr, "  FDES: input file %s: unknown extension\n", input_name);
        return FDES_EINVAL;
    }
    fdes_params p0;
    int rc = fdes_params_init(&p0, 1000); // allocParams(&params0, 1000), src/paramStructure.cu:604-606
    if (rc) return rc;
    fdes_atoms atoms = {0, nullptr, nullptr, nullptr, nullptr};
    const bool external = atomsArray != nullptr; // atomsFromExternal, src/FDESExport.cu:73
    // the file's own atoms are read even when the caller brings a list: the readers echo them (below)
    int flags = FDES_CNF_BUG_COMPATIBLE;
    if (std::getenv("FDES_STRICT_CNF")) flags &= ~FDES_CNF_BUG_COMPATIBLE;
    rc = is_emd   ? fdes_read_emd(input_name, &p0, &atoms, flags)
         : is_qsc ? fdes_read_qsc(input_name, &p0, &atoms, flags)
                  : fdes_read_cnf(input_name, &p0, &atoms, flags);
    if (rc) {
        std::fprintf(stderr, "  FDES: cannot read simulation configuration from %s (%d)\n", input_name, rc);
        fdes_params_release(&p0);
        return rc;
    }
    // Parameter echoes happen inside the reference's readers, i.e. with the file's own atoms and after consitentParams:
    // dataFDES_used.cnf (src/paramStructure.cu:629-631), ParamsUsedQsc.txt (src/rwQsc.cu:1097), ParamsUsedEmd.txt
    // (src/rwHdf5.cu:2561-2565); a caller-supplied list is echoed as testRead.txt (src/paramStructure.cu:344).
    rc = fdes_params_consistent(&p0);
    if (rc == FDES_OK)
        fdes_write_cnf(is_emd ? "ParamsUsedEmd.txt" : is_qsc ? "ParamsUsedQsc.txt" : "dataFDES_used.cnf", &p0, &atoms);
    if (rc == FDES_OK && external) {
        rc = fdes_atoms_from_array(&atoms, atomsArray, numAtoms, /*truncate_occ=*/1); // src/paramStructure.cu:323
        if (rc) { fdes_atoms_release(&atoms); fdes_params_release(&p0); return rc; }
        p0.nAt = numAtoms;
        fdes_write_cnf("testRead.txt", &p0, &atoms);
    }
    std::fprintf(stderr, "  Number of atoms in the specimen: %i\n", atoms.nAt);
    if (rc == FDES_OK && !is_emd)
        (void)fdes_write_emd("config.emd", &p0, &atoms, nullptr, nullptr, nullptr, 0); // confOption != 0: src/FDES.cu:229-232, FDESExport.cu:149-152
    std::vector<float> image, potential, exitwave;
    fdes_ctx* ctx = nullptr;
    // Extension: FDES_DEVICES="0,1,2,3" (or FDES_NUM_GPUS=n: devices gpu_index ... gpu_index + n - 1) spreads the
    // (measurement, phonon configuration) pairs over several GPUs (every print_level).
    std::vector<int> devices;
    if (const char* e = std::getenv("FDES_DEVICES")) {
        for (const char* q = e; *q;) {
            char* end = nullptr;
            long v = std::strtol(q, &end, 10);
            if (end == q) break;
            devices.push_back((int)v);
            q = (*end == ',') ? end + 1 : end;
        }
    } else if (const char* e = std::getenv("FDES_NUM_GPUS")) {
        for (int i = 0; i < std::atoi(e); i++) devices.push_back(gpu_index + i);
    }
    if (rc == FDES_OK)
        std::fprintf(stderr, "  Wave %d x %d, slice loop: %s\n", p0.m1, p0.m2,
                     fdes_grid_backend(p0.m1, p0.m2, 0) == 2 ? "fused LDS passes" : "rocFFT + point-wise kernels");
    if (rc == FDES_OK && devices.size() > 1) {
        const size_t m12 = (size_t)p0.m1 * p0.m2;
        image.resize((size_t)p0.n1 * p0.n2 * p0.n3);
        if (print_level > 0) potential.resize(2 * m12 * (size_t)p0.m3);
        if (print_level > 1) exitwave.resize(2 * m12 * (size_t)p0.n3);
        std::fprintf(stderr, "  FDES: %zu GPUs\n", devices.size());
        rc = fdes_build_measurements_multi((int)devices.size(), devices.data(), &p0, &atoms, image.data(),
                                           print_level > 0 ? potential.data() : nullptr, print_level > 1 ? exitwave.data() : nullptr);
        if (rc) std::fprintf(stderr, "  FDES: multi-GPU simulation failed (%d)\n", rc);
    } else {
    if (rc == FDES_OK) {
        rc = fdes_create(&ctx, gpu_index);
        if (rc) std::fprintf(stderr, "  FDES: no usable GPU with index %d\n", gpu_index);
        // progressCounter, src/optimFunctions.cu:257: a percentage on stderr (FDES_QUIET=1 turns it off)
        else if (!std::getenv("FDES_QUIET")) fdes_set_progress(ctx, print_progress, nullptr, 250);
    }
    if (rc == FDES_OK) {
        const size_t m12 = (size_t)p0.m1 * p0.m2;
        image.resize((size_t)p0.n1 * p0.n2 * p0.n3);
        if (print_level > 0) potential.resize(2 * m12 * (size_t)p0.m3);
        if (print_level > 1) exitwave.resize(2 * m12 * (size_t)p0.n3);
        rc = fdes_build_measurements(ctx, &p0, &atoms, image.data(), print_level > 0 ? potential.data() : nullptr,
                                     print_level > 1 ? exitwave.data() : nullptr);
        if (rc) std::fprintf(stderr, "  FDES: simulation failed: %s\n", fdes_last_error(ctx));
    }
    }
    if (rc == FDES_OK) {
        if (image_name) rc = fdes_write_binary(image_name, image.data(), image.size()); // src/crystalMaker.cu:399
        if (rc == FDES_OK && emd_name) {
            int e = fdes_write_emd(emd_name, &p0, &atoms, image.data(), print_level > 0 ? potential.data() : nullptr,
                                   print_level > 1 ? exitwave.data() : nullptr, print_level); // :402
            if (e == FDES_EUNSUPPORTED) std::fprintf(stderr, "  FDES: libhdf5 not available, %s not written\n", emd_name);
            else if (e) rc = e;
        }
        if (dstImage) std::memcpy(dstImage, image.data(), sizeof(float) * image.size()); // exportFormedimage
    }
    if (ctx) fdes_destroy(ctx);
    fdes_atoms_release(&atoms);
    fdes_params_release(&p0);
    return rc;
}

// Same symbol, same arguments as src/FDESExport.cu:59-60.  Never exits the host process.
extern "C" void FDES(int gpu_Index, int print_Level, char* input_name, char* image_name, char* emd_save_name,
                     float* atomsArray, int numAtoms, float* dstImage)
{
    std::fprintf(stderr, "   input_name %s  \n", input_name ? input_name : "(null)");
    int rc = fdes_run_file(gpu_Index, print_Level, input_name, image_name, emd_save_name, atomsArray, numAtoms, dstImage);
    if (rc) std::fprintf(stderr, "  FDES: failed with code %d\n", rc);
}
