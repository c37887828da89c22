// parsers_asan.cpp — AddressSanitizer / UBSan harness for the host-side readers and writers (cnf.cpp, qsc.cpp,
// emd.cpp, params.cpp; no GPU code).  TEST INFRASTRUCTURE.  Reads the reference's shipped inputs (tests/golden) in
// every reader mode, round-trips them through the writers, and then feeds the readers damaged inputs: truncated
// files, over-long lines, missing values, garbage, an atom count that does not match.  The readers must either parse
// or return an error code; the sanitizers catch everything else.
// Build + run: tests/test_host_cpu.py::test_host_parsers_under_address_sanitizer
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/fdes_abi.h"

static std::string slurp(const std::string& f)
{
    std::ifstream in(f, std::ios::binary);
    std::stringstream ss;
    ss << in.rdbuf();
    return ss.str();
}
static void spit(const std::string& f, const std::string& s)
{
    std::ofstream out(f, std::ios::binary);
    out << s;
}

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("EXPECT failed: %s (line %d)\n", #c, __LINE__); fails++; } } while (0)

typedef int (*reader_fn)(const char*, fdes_params*, fdes_atoms*, int);

static int read_with(reader_fn fn, const std::string& file, int flags, int* nat = nullptr, fdes_params* keep = nullptr)
{
    fdes_params p;
    if (fdes_params_init(&p, 1000)) return -100;
    fdes_atoms a = {0, nullptr, nullptr, nullptr, nullptr};
    int rc = fn(file.c_str(), &p, &a, flags);
    if (rc == FDES_OK) rc = fdes_params_consistent(&p) ? -101 : FDES_OK;
    if (rc == FDES_OK && !(flags & FDES_CNF_SKIP_ATOMS)) {
        // touch every atom: a reader that over-reports nAt is caught here
        double s = 0;
        for (int i = 0; i < a.nAt; i++) s += a.Z[i] + a.xyz[3 * i] + a.xyz[3 * i + 1] + a.xyz[3 * i + 2] + a.dwf[i] + a.occ[i];
        if (s != s) std::printf("(nan in atoms of %s)\n", file.c_str());
    }
    if (nat) *nat = a.nAt;
    if (keep && rc == FDES_OK) { *keep = p; keep->tiltspec = keep->tiltbeam = keep->defoci = nullptr; }
    fdes_atoms_release(&a);
    fdes_params_release(&p);
    return rc;
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::printf("usage: parsers_asan <golden dir> <scratch dir>\n"); return 2; }
    const std::string G = argv[1], T = argv[2];
    // ---- the shipped inputs, every mode
    for (const char* f : {"dataFDES_bin.cnf", "dataFDES_Auparticle.cnf", "dataFDES_Si001_11k.cnf"}) {
        int n_clean = 0, n_bug = 0;
        EXPECT(read_with(fdes_read_cnf, G + "/" + f, 0, &n_clean) == FDES_OK);
        EXPECT(read_with(fdes_read_cnf, G + "/" + f, FDES_CNF_BUG_COMPATIBLE, &n_bug) == FDES_OK);
        EXPECT(read_with(fdes_read_cnf, G + "/" + f, FDES_CNF_SKIP_ATOMS) == FDES_OK);
        EXPECT(n_clean > 0 && (n_bug == n_clean || n_bug == n_clean + 1)); // duplicated-last-atom quirk
        std::printf("%s: %d atoms (%d bug-compatible)\n", f, n_clean, n_bug);
    }
    int nq = 0;
    EXPECT(read_with(fdes_read_qsc, G + "/qsc/test.qsc", 0, &nq) == FDES_OK);
    EXPECT(nq == 8100);
    // ---- write -> read round trip
    {
        fdes_params p;
        fdes_params_init(&p, 1000);
        fdes_atoms a = {0, nullptr, nullptr, nullptr, nullptr};
        EXPECT(fdes_read_cnf((G + "/dataFDES_Auparticle.cnf").c_str(), &p, &a, 0) == FDES_OK);
        EXPECT(fdes_params_consistent(&p) == FDES_OK);
        EXPECT(fdes_write_cnf((T + "/round.cnf").c_str(), &p, &a) == FDES_OK);
        int n2 = 0;
        EXPECT(read_with(fdes_read_cnf, T + "/round.cnf", 0, &n2) == FDES_OK && n2 == a.nAt);
        if (fdes_emd_available()) {
            std::vector<float> img((size_t)p.n1 * p.n2 * p.n3, 0.5f);
            EXPECT(fdes_write_emd((T + "/round.emd").c_str(), &p, &a, img.data(), nullptr, nullptr, 0) == FDES_OK);
            int n3 = 0;
            EXPECT(read_with(fdes_read_emd, T + "/round.emd", 0, &n3) == FDES_OK && n3 == a.nAt);
            EXPECT(read_with(fdes_read_emd, G + "/Auparticle_config.emd", 0, &n3) == FDES_OK && n3 == 309);
            spit(T + "/garbage.emd", "not an hdf5 file at all");
            EXPECT(read_with(fdes_read_emd, T + "/garbage.emd", 0) != FDES_OK);
        } else {
            std::printf("(libhdf5 not available: .emd legs skipped)\n");
        }
        fdes_atoms_release(&a);
        fdes_params_release(&p);
    }
    // ---- damaged .cnf inputs: any return code is fine, memory errors are not
    const std::string good = slurp(G + "/dataFDES_Auparticle.cnf");
    std::vector<std::string> bad;
    bad.push_back("");
    bad.push_back("\n\n\n");
    bad.push_back("atom:");
    bad.push_back("atom: 79\n");
    bad.push_back("atom: 79 1e-10 2e-10\natom:\natom: x y z\n");
    bad.push_back("image_size_z: 5000\nspecimen_tilt: 1 2\n" + std::string(3000, 'x') + "\n");
    bad.push_back("image_size_z: -3\nimage_size_x: 0\nsample_size_z: -1\n");
    bad.push_back("image_size_z: 2\nspecimen_tilt:\n\n\n\n\nbeam_tilt: 1\n\ndefoci:\n\n");
    bad.push_back("user_name: " + std::string(5000, 'u') + "\ncomment: " + std::string(2000, 'c'));
    bad.push_back(std::string(99, 'a') + ": 1\n" + std::string(100, 'b') + ": 2\n" + std::string(101, 'c') + ": 3\n");
    bad.push_back("voltage: 1e400\npixel_size_x: nan\npixel_size_y: inf\nmode: 99999999999999999999\n");
    for (size_t cut : {(size_t)1, (size_t)17, good.size() / 3, good.size() / 2, good.size() - 1, good.size() - 7}) bad.push_back(good.substr(0, cut));
    { std::string g = good; for (size_t i = 0; i < g.size(); i += 97) g[i] = '\0'; bad.push_back(g); }
    { std::string g = good; for (size_t i = 0; i < g.size(); i += 53) g[i] = ':'; bad.push_back(g); }
    int k = 0;
    for (const auto& b : bad) {
        const std::string f = T + "/bad" + std::to_string(k++) + ".cnf";
        spit(f, b);
        for (int flags : {0, FDES_CNF_BUG_COMPATIBLE, FDES_CNF_SKIP_ATOMS}) (void)read_with(fdes_read_cnf, f, flags);
    }
    EXPECT(read_with(fdes_read_cnf, T + "/does_not_exist.cnf", 0) != FDES_OK);
    // ---- damaged .qsc / .cfg inputs
    const std::string qsc = slurp(G + "/qsc/test.qsc"), cfg = slurp(G + "/qsc/SrTiO3.cfg");
    spit(T + "/SrTiO3.cfg", cfg);
    spit(T + "/ok.qsc", qsc);
    EXPECT(read_with(fdes_read_qsc, T + "/ok.qsc", 0, &nq) == FDES_OK && nq == 8100);
    std::vector<std::string> badq;
    badq.push_back("");
    badq.push_back("mode: TEM\n");
    badq.push_back("mode: STEM\nfilename: SrTiO3.cfg\n");
    badq.push_back("mode: TEM\nfilename: missing.cfg\nnx: 64\n");
    badq.push_back("mode: TEM\nfilename: SrTiO3.cfg\nNCELLX: -3\nNCELLY: 0\nNCELLZ: 999999999\nnx: 64\n");
    badq.push_back("mode: TEM\nfilename: " + std::string(4000, 'f') + "\n");
    for (size_t cut : {qsc.size() / 4, qsc.size() / 2, qsc.size() - 3}) badq.push_back(qsc.substr(0, cut));
    k = 0;
    for (const auto& b : badq) {
        const std::string f = T + "/badq" + std::to_string(k++) + ".qsc";
        spit(f, b);
        (void)read_with(fdes_read_qsc, f, 0);
    }
    for (size_t cut : {(size_t)0, (size_t)10, cfg.size() / 3, cfg.size() / 2, cfg.size() - 2}) { // damaged cell files
        spit(T + "/SrTiO3.cfg", cfg.substr(0, cut));
        (void)read_with(fdes_read_qsc, T + "/ok.qsc", 0);
    }
    { std::string c2 = cfg; for (size_t i = 0; i < c2.size(); i += 31) c2[i] = 'Q'; spit(T + "/SrTiO3.cfg", c2); (void)read_with(fdes_read_qsc, T + "/ok.qsc", 0); }
    // ---- seeded mutation fuzz: byte flips, deleted / duplicated / swapped lines, numbers replaced by extremes.  400 .cnf and
    // 200 .qsc mutants, all reader modes; the only requirement is that nothing but a return code comes back.
    {
        unsigned long long st = 0x9E3779B97F4A7C15ull;
        auto rnd = [&](size_t n) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return n ? (size_t)(st % n) : (size_t)0; };
        auto lines_of = [](const std::string& t) {
            std::vector<std::string> L;
            std::stringstream ss(t);
            std::string l;
            while (std::getline(ss, l)) L.push_back(l);
            return L;
        };
        const char* extremes[] = {"-1", "0", "1e38", "-1e38", "1e-45", "nan", "inf", "2147483648", "-2147483649", "99999999999999999999", "", "0x10", "1,5", "1e", "--3", "+"};
        auto mutate = [&](const std::string& src) {
            std::vector<std::string> L = lines_of(src);
            const int ops = 1 + (int)rnd(4);
            for (int o = 0; o < ops && !L.empty(); o++) {
                const size_t i = rnd(L.size());
                switch (rnd(6)) {
                case 0: L.erase(L.begin() + (long)i); break;
                case 1: L.insert(L.begin() + (long)i, L[rnd(L.size())]); break;
                case 2: std::swap(L[i], L[rnd(L.size())]); break;
                case 3: { // replace the value after the colon (or the whole line) by an extreme
                    const size_t c = L[i].find(':');
                    L[i] = (c == std::string::npos ? std::string() : L[i].substr(0, c + 1) + " ") + extremes[rnd(sizeof(extremes) / sizeof(extremes[0]))];
                    break;
                }
                case 4: if (!L[i].empty()) L[i][rnd(L[i].size())] = (char)rnd(256); break;
                default: L[i] += " " + std::string(rnd(300), (char)('0' + rnd(10))); break;
                }
            }
            std::string out;
            for (const auto& l : L) out += l + "\n";
            if (rnd(8) == 0) out.resize(rnd(out.size() + 1));
            return out;
        };
        // a short .cnf (the header of the shipped one + its first 40 atoms) keeps 400 x 3 parses fast
        std::string small;
        {
            std::vector<std::string> L = lines_of(good);
            int atoms = 0;
            for (const auto& l : L) {
                const bool is_atom = l.rfind("atom:", 0) == 0;
                if (is_atom && ++atoms > 40) continue;
                small += l + "\n";
            }
        }
        for (int m = 0; m < 400; m++) {
            const std::string f = T + "/fuzz.cnf";
            spit(f, mutate(small));
            for (int flags : {0, FDES_CNF_BUG_COMPATIBLE, FDES_CNF_SKIP_ATOMS}) (void)read_with(fdes_read_cnf, f, flags);
        }
        spit(T + "/SrTiO3.cfg", cfg);
        for (int m = 0; m < 200; m++) {
            spit(T + "/fuzz.qsc", mutate(qsc));
            (void)read_with(fdes_read_qsc, T + "/fuzz.qsc", 0);
            if (m % 4 == 0) { // the cell file as well
                spit(T + "/SrTiO3.cfg", mutate(cfg));
                (void)read_with(fdes_read_qsc, T + "/ok.qsc", 0);
                spit(T + "/SrTiO3.cfg", cfg);
            }
        }
        // the same cell as .cssr and .dat (qsc.cpp read_cssr_* / read_dat_*): whole, cut short, mutated
        const std::string cssr = " 3.905 3.905 3.905\n 90 90 90 SPGR = 1 P 1 OPT = 1\n 5 0\n SrTiO3\n"
                                 "   1 Sr  0.0 0.0 0.0  0 0 0 0 0 0 0 0  0.6214\n   2 Ti1 0.5 0.5 0.5  0 0 0 0 0 0 0 0  0.4390\n"
                                 "   3 O1  0.0 0.5 0.5  0 0 0 0 0 0 0 0  0.7323\n   4 O2  0.5 0.0 0.5  0 0 0 0 0 0 0 0  0.7323\n"
                                 "   5 O3  0.5 0.5 0.0  0 0 0 0 0 0 0 0  0.7323\n";
        const std::string dat = "Number of atoms = 5\na = 3.905\nb = 3.905\nc = 3.905\nalpha = 90\nbeta = 90\ngamma = 90\n"
                                "Sr 0.0 0.0 0.0\nTi 0.5 0.5 0.5\n\nO  0.0 0.5 0.5\nO  0.5 0.0 0.5\nO  0.5 0.5 0.0\n";
        const std::string base = "mode: TEM\nNCELLX: 2\nNCELLY: 2\nNCELLZ: 2\nslices: 4\nnx: 32\nv0: 80\nCs: 1.2\nalpha: 0.5\n";
        spit(T + "/c.qsc", base + "filename: cell.cssr\n");
        spit(T + "/d.qsc", base + "filename: cell.dat\n");
        spit(T + "/cell.cssr", cssr);
        spit(T + "/cell.dat", dat);
        int nc = 0, nd = 0;
        EXPECT(read_with(fdes_read_qsc, T + "/c.qsc", 0, &nc) == FDES_OK && nc == 40);
        EXPECT(read_with(fdes_read_qsc, T + "/d.qsc", 0, &nd) == FDES_OK && nd == 40);
        for (size_t cut : {(size_t)0, (size_t)5, (size_t)25, cssr.size() / 2, cssr.size() - 8}) {
            spit(T + "/cell.cssr", cssr.substr(0, cut));
            EXPECT(read_with(fdes_read_qsc, T + "/c.qsc", 0) != FDES_OK);
        }
        for (size_t cut : {(size_t)0, (size_t)12, dat.size() / 2, dat.size() - 20}) {
            spit(T + "/cell.dat", dat.substr(0, cut));
            EXPECT(read_with(fdes_read_qsc, T + "/d.qsc", 0) != FDES_OK);
        }
        for (int m = 0; m < 50; m++) {
            spit(T + "/cell.cssr", mutate(cssr));
            (void)read_with(fdes_read_qsc, T + "/c.qsc", 0);
            spit(T + "/cell.dat", mutate(dat));
            (void)read_with(fdes_read_qsc, T + "/d.qsc", 0);
        }
        // boxed mode and a cell with partial / shared occupancy: whole and mutated
        {
            std::string occ = cfg;
            const std::string full = "0.5 0.5 0.5  0.4390  1.0";
            const size_t at = occ.find(full);
            if (at != std::string::npos) occ.replace(at, full.size(), "0.5 0.5 0.5  0.4390  0.5");
            occ += "14\nN\n0.5 0.5 0.5  0.5 0.3\n";
            const size_t np5 = occ.find("Number of particles = 5");
            if (np5 != std::string::npos) occ.replace(np5, 23, "Number of particles = 6");
            spit(T + "/SrTiO3.cfg", occ);
            spit(T + "/b.qsc", base + "filename: SrTiO3.cfg\nCube: 13 11 9.5\nCrystal tilt Z: 0.2\nxOffset: 1.5\ntds: yes\ntemperature: 600\n");
            int nb = 0;
            EXPECT(read_with(fdes_read_qsc, T + "/b.qsc", 0, &nb) == FDES_OK && nb > 20);
            EXPECT(read_with(fdes_read_qsc, T + "/ok.qsc", 0, &nb) == FDES_OK && nb == 6 * 1620);
            for (int m = 0; m < 50; m++) {
                spit(T + "/fuzzb.qsc", mutate(base + "filename: SrTiO3.cfg\nCube: 13 11 9.5\nCrystal tilt Z: 0.2\nxOffset: 1.5\n"));
                (void)read_with(fdes_read_qsc, T + "/fuzzb.qsc", 0);
                spit(T + "/SrTiO3.cfg", mutate(occ));
                (void)read_with(fdes_read_qsc, T + "/b.qsc", 0);
                spit(T + "/SrTiO3.cfg", occ);
            }
            spit(T + "/SrTiO3.cfg", cfg);
        }
        std::printf("fuzz: 400 .cnf, 200 .qsc, 50 .cfg, 50 .cssr, 50 .dat, 100 boxed / vacancy mutants parsed or refused\n");
    }
    // ---- atoms from a flat array (the legacy export's path)
    {
        std::vector<float> arr = {79, 0, 0, 0, 6e-21f, 1.7f, 14, 1e-10f, 0, 0, 6e-21f, 0.4f};
        fdes_atoms a = {0, nullptr, nullptr, nullptr, nullptr};
        EXPECT(fdes_atoms_from_array(&a, arr.data(), 2, 1) == FDES_OK && a.nAt == 2 && a.occ[0] == 1.f && a.occ[1] == 0.f);
        fdes_atoms_release(&a);
        (void)fdes_atoms_from_array(&a, arr.data(), 0, 0); // empty list: either answer, no memory error
        fdes_atoms_release(&a);
    }
    std::printf(fails ? "FAILED (%d)\n" : "all ok\n", fails);
    return fails ? 1 : 0;
}
