// qsc.cpp — QSTEM `.qsc` front-end: fills fdes_params + the host atom list the way the
// reference's readQsc does (src/rwQsc.cu:8-1101), including the `.cfg` unit-cell reader and
// NCELL replication it borrows from qstem-libs (readparams.cpp:173-212,
// fileio_fftw3.cpp:721-779, 908-989, 1188-1306, 1313-1657).  Host C++ only.
//
// Behaviour kept on purpose (bug-for-bug, so that a .qsc that runs under the reference gives
// the same params_t here):
//   * `title` lookup is a substring match on the comment-stripped line ('%' only — the
//     reference's setComment('#') has no effect, readparams.cpp:184), starting at the current
//     file position and wrapping once; "mode: STEM" therefore passes as TEM (strstr "TEM").
//   * crystal tilt is applied to the super cell about its centre AND handed on as
//     specimen_tilt_offset (rwQsc.cu:954-956); a trailing "deg" converts deg->rad although
//     the bare number is documented as mrad (rwQsc.cu:98-138).
//   * dn = round(n/2) with integer division, m3 = slices, subSlTh = d3/10, C5_0 = C5[A]*1e-3,
//     A1_0 = astigmatism[A]*1e-9, A1_1 = angle[rad]*1e-9, mtf_d is never read (rwQsc.cu:943-1012).
//   * atoms are shifted by -(max-min)/2 with min starting at 1 and max at 0 (rwQsc.cu:1041-1083).
// `tds: yes` applies QSTEM's Einstein displacements at read time (the reference seeds them from the clock; here a fixed
// seed - FDES's own frozen phonons are the `frozen_phonons` of the engine and independent of this).  `Cube:` boxes
// the crystal the way tiltBoxed does.  Unit cells
// with partial or shared site occupancy draw their vacancies with ran1() from its fixed seed, as QSTEM does.
// `.cssr` and `.dat` cells are read the way the vendored readUnitCell reads them; the reference's
// own readQsc ends the program for any cell file whose name holds no ".cfg" (rwQsc.cu:976-983).
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fdes_internal.h"

namespace {

constexpr size_t kParBuf = 1024; // PAR_BUF_LEN, readparams.cpp:40

// The key:value file of readparams.cpp, kept as fgets() would deliver it (pieces of at most
// kParBuf-1 bytes, newline retained) with a read cursor.
struct ParFile {
    std::vector<std::string> lines;
    size_t pos = 0;

    bool open(const char* name)
    {
        FILE* f = std::fopen(name, "r");
        if (!f) return false;
        char buf[kParBuf];
        while (std::fgets(buf, (int)kParBuf, f)) lines.emplace_back(buf);
        std::fclose(f);
        pos = 0;
        return true;
    }
    void rewind() { pos = 0; }
    bool next_line(std::string& out)
    {
        if (pos >= lines.size()) return false;
        out = lines[pos++];
        return true;
    }
    // readparam(title, parString, wrapFlag), readparams.cpp:173-212
    bool find(const char* title, std::string& rest, bool wrap = true)
    {
        for (int pass = 0; pass < (wrap ? 2 : 1); pass++) {
            while (pos < lines.size()) {
                std::string l = lines[pos++];
                size_t c = l.find('%');
                if (c != std::string::npos) l.resize(c);
                size_t t = l.find(title);
                if (t != std::string::npos) {
                    rest = l.substr(t + std::strlen(title));
                    return true;
                }
            }
            if (pass == 0 && wrap) pos = 0;
        }
        return false;
    }
};

// strnext(str, " \t"), readparams.cpp:227-243: next word, NULL at end of string or line
const char* strnext(const char* s)
{
    bool found = false;
    const char* q = s;
    for (; *q; q++) {
        const bool d = (*q == ' ' || *q == '\t');
        if (d) found = true;
        if (found && !d) break;
    }
    if (*q == '\0' || *q == '\n') return nullptr;
    return q;
}

const char* const kSymbols[103] = {
    "H",  "He", "Li", "Be", "B",  "C",  "N",  "O",  "F",  "Ne", "Na", "Mg", "Al", "Si", "P",  "S",  "Cl", "Ar", "K",  "Ca", "Sc",
    "Ti", "V",  "Cr", "Mn", "Fe", "Co", "Ni", "Cu", "Zn", "Ga", "Ge", "As", "Se", "Br", "Kr", "Rb", "Sr", "Y",  "Zr", "Nb", "Mo",
    "Tc", "Ru", "Rh", "Pd", "Ag", "Cd", "In", "Sn", "Sb", "Te", "I",  "Xe", "Cs", "Ba", "La", "Ce", "Pr", "Nd", "Pm", "Sm", "Eu",
    "Gd", "Tb", "Dy", "Ho", "Er", "Tm", "Yb", "Lu", "Hf", "Ta", "W",  "Re", "Os", "Ir", "Pt", "Au", "Hg", "Tl", "Pb", "Bi", "Po",
    "At", "Rn", "Fr", "Ra", "Ac", "Th", "Pa", "U",  "Np", "Pu", "Am", "Cm", "Bk", "Cf", "Es", "Fm", "Md", "No", "Lr"};

// getZNumber, fileio_fftw3.cpp:2299-2321: the first two characters of the line, a lone letter
// padded with a blank, searched as a substring of the concatenated two-character symbol table.
int z_number(const std::string& line)
{
    char e[3] = {line.size() > 0 ? line[0] : '\0', line.size() > 1 ? line[1] : '\0', '\0'};
    if (e[0] == '\0') return 0;
    if (std::atoi(e + 1) != 0 || e[1] == '\n' || e[1] == '\0' || e[1] == '\r') e[1] = ' ';
    static const std::string table = [] { // initialised once, thread-safe (C++11 static)
        std::string t;
        for (const char* s : kSymbols) {
            t += s;
            if (std::strlen(s) == 1) t += ' ';
        }
        return t;
    }();
    size_t at = table.find(e);
    return at == std::string::npos ? 0 : (int)(at / 2) + 1;
}

struct QAtom { // atomStruct, stemtypes_fftw3.h:70-77 (floats, as there)
    float x, y, z, dw, occ;
    int Znum;
};

struct Cell {
    double Mm[3][3];
    float ax, by, c;
    int ncoord;
};

int fail(int code, const char* what, const char* file)
{
    std::fprintf(stderr, "  FDES(.qsc): %s%s%s\n", what, file ? ": " : "", file ? file : "");
    return code;
}

// readCFGCellParams, fileio_fftw3.cpp:721-779
int read_cfg_cell(const char* file, Cell& cell)
{
    ParFile f;
    if (!f.open(file)) return fail(FDES_EIO, "could not open CFG input file", file);
    std::string r;
    int ncoord = 0;
    double scale = 1.0;
    if (f.find("Number of particles =", r)) std::sscanf(r.c_str(), "%d", &ncoord);
    if (f.find("A =", r)) std::sscanf(r.c_str(), "%lf", &scale);
    std::memset(cell.Mm, 0, sizeof(cell.Mm));
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            char key[32];
            std::snprintf(key, sizeof(key), "H0(%d,%d) =", a + 1, b + 1);
            if (f.find(key, r)) std::sscanf(r.c_str(), "%lf", &cell.Mm[a][b]);
        }
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) cell.Mm[a][b] *= scale;
    auto len = [&](int a) { return std::sqrt(cell.Mm[a][0] * cell.Mm[a][0] + cell.Mm[a][1] * cell.Mm[a][1] + cell.Mm[a][2] * cell.Mm[a][2]); };
    cell.ax = (float)len(0);
    cell.by = (float)len(1);
    cell.c = (float)len(2);
    cell.ncoord = ncoord;
    if (ncoord < 1) return fail(FDES_EINVAL, "number of atoms in CFG file not specified", file);
    return FDES_OK;
}

// readNextCFGAtom called ncoord times, fileio_fftw3.cpp:908-989; atoms are stored from the
// back of the array forwards (fileio_fftw3.cpp:1431-1438)
int read_cfg_atoms(const char* file, int ncoord, std::vector<QAtom>& atoms)
{
    ParFile f;
    if (!f.open(file)) return fail(FDES_EIO, "could not open CFG input file", file);
    std::string r;
    const bool noVelocity = f.find(".NO_VELOCITY.", r);
    int entryCount = 3;
    if (f.find("entry_count =", r)) std::sscanf(r.c_str(), "%d", &entryCount);
    if (!noVelocity) entryCount += 3;
    if (entryCount < 3 || entryCount > 64) return fail(FDES_EINVAL, "bad entry_count in", file);
    const int off = 3 * (noVelocity ? 0 : 1);
    std::vector<double> data((size_t)entryCount + 1, 0.0);
    double mass = 28;
    int element = 1;
    atoms.assign((size_t)ncoord, QAtom{});
    std::string buf;
    for (int i = ncoord - 1; i >= 0; i--) {
        if (!f.next_line(buf)) return fail(FDES_EINVAL, "number of atoms does not agree with atoms in file", file);
        const char* nx = strnext(buf.c_str());
        if (std::atof(buf.c_str()) >= 1.0 && (nx == nullptr || *nx == '#')) { // mass line, then the symbol, then data
            mass = std::atof(buf.c_str());
            if (!f.next_line(buf)) return fail(FDES_EINVAL, "number of atoms does not agree with atoms in file", file);
            element = z_number(buf);
            if (!f.next_line(buf)) return fail(FDES_EINVAL, "number of atoms does not agree with atoms in file", file);
        }
        const char* s = buf.c_str();
        while (*s == ' ' || *s == '\t') s++;
        for (int j = 0; j < entryCount; j++) {
            if (!s) return fail(FDES_EINVAL, "incomplete atom data line in", file);
            data[(size_t)j] = std::atof(s);
            s = strnext(s);
        }
        QAtom& a = atoms[(size_t)i];
        a.Znum = element;
        a.x = (float)data[0];
        a.y = (float)data[1];
        a.z = (float)data[2];
        a.dw = (float)(0.45 * 28.0 / mass);
        a.occ = 1.0f;
        if (entryCount > 3 + off) a.dw = (float)data[(size_t)(3 + off)];
        if (entryCount > 4 + off) a.occ = (float)data[(size_t)(4 + off)];
        if (a.Znum < 1 || a.Znum > 103) return fail(FDES_EINVAL, "bad atomic number in", file);
    }
    return FDES_OK;
}

// makeCellVectMuls, matrixlib.cpp:722-747.  The lattice parameters and angles are floats in
// the reference (data_containers.h:161-162).  Kept as written there, including vby[1] =
// by*cos(gamma) (not sin) for a non-orthogonal cell.
void cell_vectors(Cell& cell, float a, float b, float c, float alpha, float beta, float gamma)
{
    const double d = 1.7453292519943e-2; // PI180, matrixlib.h:27
    std::memset(cell.Mm, 0, sizeof(cell.Mm));
    cell.ax = a;
    cell.by = b;
    cell.c = c;
    cell.Mm[0][0] = a;
    if (alpha == 90 && beta == 90 && gamma == 90) {
        cell.Mm[1][1] = b;
        cell.Mm[2][2] = c;
        return;
    }
    const double ca = std::cos(alpha * d), cb = std::cos(beta * d), cg = std::cos(gamma * d), sg = std::sin(gamma * d);
    cell.Mm[1][0] = b * cg;
    cell.Mm[1][1] = b * cg;
    cell.Mm[2][0] = c * cb;
    cell.Mm[2][1] = c * (ca - cb * cg) / sg;
    cell.Mm[2][2] = c * (std::sqrt(1 - ca * ca - cb * cb + 2 * ca * cb * cg) / sg);
}

// readCSSRCellParams, fileio_fftw3.cpp:785-815: "a b c" / "alpha beta gamma SPGR = 1 ..." /
// atom count.  A missing "SPGR =" crashes the reference; here it is an error.
int read_cssr_cell(const char* file, Cell& cell)
{
    ParFile f;
    if (!f.open(file)) return fail(FDES_EIO, "could not open CSSR input file", file);
    std::string l1, l2, l3;
    if (!f.next_line(l1) || !f.next_line(l2) || !f.next_line(l3)) return fail(FDES_EINVAL, "CSSR header is incomplete", file);
    char s1[64] = "", s2[64] = "", s3[64] = "";
    std::sscanf(l1.c_str(), " %63s %63s %63s", s1, s2, s3);
    const float a = (float)std::atof(s1), b = (float)std::atof(s2), c = (float)std::atof(s3);
    s1[0] = s2[0] = s3[0] = '\0';
    std::sscanf(l2.c_str(), " %63s %63s %63s", s1, s2, s3);
    const float alpha = (float)std::atof(s1), beta = (float)std::atof(s2), gamma = (float)std::atof(s3);
    cell_vectors(cell, a, b, c, alpha, beta, gamma);
    const size_t sp = l2.find("SPGR =");
    if (sp == std::string::npos) return fail(FDES_EINVAL, "no 'SPGR =' on the second line of", file);
    if (std::atoi(l2.c_str() + sp + 6) != 1) return fail(FDES_EUNSUPPORTED, "cannot interpret a space group other than 1 in", file);
    cell.ncoord = std::atoi(l3.c_str());
    if (cell.ncoord < 1) return fail(FDES_EINVAL, "number of atoms in CSSR file not specified", file);
    return FDES_OK;
}

// readNextCSSRAtom called ncoord times, fileio_fftw3.cpp:998-1052: four header lines, then
// "index element x y z  c1..c8  dw" per atom.  The reference leaves dw uninitialised when the
// line is short; here a short line is an error.
int read_cssr_atoms(const char* file, int ncoord, std::vector<QAtom>& atoms)
{
    ParFile f;
    if (!f.open(file)) return fail(FDES_EIO, "could not open CSSR input file", file);
    std::string buf;
    for (int i = 0; i < 4; i++)
        if (!f.next_line(buf)) return fail(FDES_EINVAL, "CSSR header is incomplete", file);
    atoms.assign((size_t)ncoord, QAtom{});
    for (int i = ncoord - 1; i >= 0; i--) {
        if (!f.next_line(buf)) return fail(FDES_EINVAL, "number of atoms does not agree with atoms in file", file);
        int k[9];
        char el[64] = "", s1[64] = "", s2[64] = "", s3[64] = "";
        double dw = 0;
        const int got = std::sscanf(buf.c_str(), "%d %63s %63s %63s %63s %d %d %d %d %d %d %d %d %lf", &k[0], el, s1, s2, s3, &k[1], &k[2],
                                    &k[3], &k[4], &k[5], &k[6], &k[7], &k[8], &dw);
        if (got != 14) return fail(FDES_EINVAL, "incomplete atom data line in", file);
        QAtom& a = atoms[(size_t)i];
        a.x = (float)std::atof(s1);
        a.y = (float)std::atof(s2);
        a.z = (float)std::atof(s3);
        a.occ = 1.0f;
        a.Znum = z_number(el);
        a.dw = (float)dw;
        if (a.Znum < 1 || a.Znum > 103) return fail(FDES_EINVAL, "bad atomic number in", file);
    }
    return FDES_OK;
}

// readDATCellParams, fileio_fftw3.cpp:666-713.  Kept: the titles are substring matches from the
// current line on, wrapping once ("a =" is also found inside "alpha =", "beta =", "gamma ="),
// and alpha and gamma are stored swapped (:703-705).  pos_after_gamma is the line after the one
// the "gamma =" lookup of readNextDATAtom (:854) stops on, or the end of the file.
int read_dat_cell(const char* file, Cell& cell)
{
    ParFile f;
    if (!f.open(file)) return fail(FDES_EIO, "could not open DAT input file", file);
    std::string r;
    int ncoord = 0;
    double a = 0, b = 0, c = 0, alpha = 90.0, beta = 90.0, gamma = 90.0;
    if (f.find("Number of atoms =", r)) std::sscanf(r.c_str(), "%d", &ncoord);
    if (f.find("a =", r)) std::sscanf(r.c_str(), "%lf", &a);
    if (f.find("b =", r)) std::sscanf(r.c_str(), "%lf", &b);
    if (f.find("c =", r)) std::sscanf(r.c_str(), "%lf", &c);
    if (f.find("alpha =", r)) std::sscanf(r.c_str(), "%lf", &alpha);
    if (f.find("beta =", r)) std::sscanf(r.c_str(), "%lf", &beta);
    if (f.find("gamma =", r)) std::sscanf(r.c_str(), "%lf", &gamma);
    if (!(a > 0) || !(b > 0) || !(c > 0)) return fail(FDES_EINVAL, "lattice parameters a, b, c not specified in", file);
    cell_vectors(cell, (float)a, (float)b, (float)c, /*cAlpha=*/(float)gamma, (float)beta, /*cGamma=*/(float)alpha);
    cell.ncoord = ncoord;
    if (ncoord < 1) return fail(FDES_EINVAL, "number of atoms in DAT file not specified", file);
    return FDES_OK;
}

// readNextDATAtom called ncoord times, fileio_fftw3.cpp:825-895: after the "gamma =" line, every
// line whose first two characters name an element is an atom: "El x y z"; dw = 0.45*28/(2 Z).
int read_dat_atoms(const char* file, int ncoord, std::vector<QAtom>& atoms)
{
    ParFile f;
    if (!f.open(file)) return fail(FDES_EIO, "could not open DAT input file", file);
    std::string buf;
    f.find("gamma =", buf);
    atoms.assign((size_t)ncoord, QAtom{});
    for (int i = ncoord - 1; i >= 0; i--) {
        int element = 0;
        do {
            if (!f.next_line(buf)) return fail(FDES_EINVAL, "number of atoms does not agree with atoms in file", file);
            element = z_number(buf);
        } while (element == 0);
        const char* s = buf.c_str() + (buf.size() < 2 ? buf.size() : 2);
        while (*s == ' ' || *s == '\t') s++;
        double data[3] = {0, 0, 0};
        for (int j = 0; j < 3; j++) {
            if (!s) return fail(FDES_EINVAL, "incomplete atom data line in", file);
            data[j] = std::atof(s);
            s = strnext(s);
        }
        QAtom& a = atoms[(size_t)i];
        a.Znum = element;
        a.x = (float)data[0];
        a.y = (float)data[1];
        a.z = (float)data[2];
        a.dw = (float)(0.45 * 28.0 / (2.0 * element));
        a.occ = 1.0f;
        if (a.Znum < 1 || a.Znum > 103) return fail(FDES_EINVAL, "bad atomic number in", file);
    }
    return FDES_OK;
}

// ran1, fileio_fftw3.cpp:2559-2599: Park-Miller minimal standard generator with Bays-Durham shuffle, seeded with -1 (the
// static idum of replicateUnitCell, :1196); uniform deviates in (0, 1)
struct Ran1 {
    static constexpr long IA = 16807, IM = 2147483647, IQ = 127773, IR = 2836, NTAB = 32, NDIV = 1 + (IM - 1) / NTAB;
    long idum = -1, iy = 0, iv[NTAB] = {};
    double next()
    {
        if (idum <= 0 || !iy) {
            idum = (-idum < 1) ? 1 : -idum;
            for (long j = NTAB + 7; j >= 0; j--) {
                const long k = idum / IQ;
                idum = IA * (idum - k * IQ) - IR * k;
                if (idum < 0) idum += IM;
                if (j < NTAB) iv[j] = idum;
            }
            iy = iv[0];
        }
        const long k = idum / IQ;
        idum = IA * (idum - k * IQ) - IR * k;
        if (idum < 0) idum += IM;
        const long j = iy / NDIV;
        iy = iv[j];
        iv[j] = idum;
        const double temp = (1.0 / IM) * iy;
        return temp > 1.0 - 1.2e-7 ? 1.0 - 1.2e-7 : temp;
    }
};

// `tds: yes`: QSTEM's Einstein displacements at read time (phononDisplacement, fileio_fftw3.cpp:367-600, Einstein branch):
// per site and cell u_i = sqrt(T / 300) sqrt(dw / (8 pi^2)) / sqrt(3) * gasdev(), cartesian, brought to reduced coordinates
// with the inverse of the (untilted) cell matrix and added to the atom's reduced position.  The reference seeds gasdev from
// the clock (:496), so there is no sequence to reproduce: the deviates here come from the same generator (gasdev over
// ran1, :2607-2640) with a fixed seed, which makes a .qsc with `tds: yes` reproducible.
struct Tds {
    bool on = false;
    double scale = 1.0;   // sqrt(tds_temp / 300), a float in the reference
    double MmInv[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    Ran1 rng;
    bool have = false;
    float gset = 0.f;
    double gasdev()
    {
        if (have) { have = false; return gset; }
        double v1, v2, rsq;
        do {
            v1 = 2.0 * rng.next() - 1.0;
            v2 = 2.0 * rng.next() - 1.0;
            rsq = v1 * v1 + v2 * v2;
        } while (rsq >= 1.0 || rsq == 0.0);
        const double fac = std::sqrt(-2.0 * std::log(rsq) / rsq);
        gset = (float)(v1 * fac);
        have = true;
        return v2 * fac;
    }
    void init(const double (*Mm)[3], double temp)
    {
        on = true;
        scale = (double)(float)std::sqrt(temp / 300.0);
        rng.idum = -20130401; // any negative seed; fixed
        double a[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) a[i * 3 + j] = Mm[j][i];
        double det = a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[6] * a[5]) + a[2] * (a[3] * a[7] - a[6] * a[4]);
        if (std::fabs(det) >= 0.0005f) { // inverse_3x3, matrixlib.cpp:225-257 (identity otherwise)
            det = 1.0f / det;
            MmInv[0] = (a[4] * a[8] - a[5] * a[7]) * det;  MmInv[1] = -(a[1] * a[8] - a[7] * a[2]) * det; MmInv[2] = (a[1] * a[5] - a[4] * a[2]) * det;
            MmInv[3] = -(a[3] * a[8] - a[5] * a[6]) * det; MmInv[4] = (a[0] * a[8] - a[6] * a[2]) * det;  MmInv[5] = -(a[0] * a[5] - a[3] * a[2]) * det;
            MmInv[6] = (a[3] * a[7] - a[6] * a[4]) * det;  MmInv[7] = -(a[0] * a[7] - a[6] * a[1]) * det; MmInv[8] = (a[0] * a[4] - a[1] * a[3]) * det;
        }
    }
    // displacement of one site in reduced coordinates (zero when off)
    void draw(double dw, double* uf)
    {
        uf[0] = uf[1] = uf[2] = 0.0;
        if (!on) return;
        const double pid = 3.14159265358979;
        const double wobble = scale * std::sqrt(dw * (1.0 / (8 * pid * pid))), sq3 = 1.0 / std::sqrt(3.0);
        double u[3];
        for (int i = 0; i < 3; i++) u[i] = wobble * sq3 * gasdev();
        for (int j = 0; j < 3; j++) // row vector times matrix, matrixProduct(&u, 1, 3, MmInv, 3, 3, &uf)
            for (int k = 0; k < 3; k++) uf[j] += u[k] * MmInv[k * 3 + j];
    }
};

// rotateVect, matrixlib.cpp:599-634 (rotation about x, then y, then z)
void rotate(double* u, double px, double py, double pz)
{
    double M[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    if (px != 0 || py != 0 || pz != 0) { // the reference's cached matrix starts as the identity
        M[0][0] = std::cos(pz) * std::cos(py);
        M[0][1] = std::cos(pz) * std::sin(py) * std::sin(px) - std::sin(pz) * std::cos(px);
        M[0][2] = std::cos(pz) * std::sin(py) * std::cos(px) + std::sin(pz) * std::sin(px);
        M[1][0] = std::sin(pz) * std::cos(py);
        M[1][1] = std::sin(pz) * std::sin(py) * std::sin(px) + std::cos(pz) * std::cos(px);
        M[1][2] = std::sin(pz) * std::sin(py) * std::cos(px) - std::cos(pz) * std::sin(px);
        M[2][0] = -std::sin(py);
        M[2][1] = std::cos(py) * std::sin(px);
        M[2][2] = std::cos(py) * std::cos(px);
    }
    double v[3];
    for (int a = 0; a < 3; a++) v[a] = M[a][0] * u[0] + M[a][1] * u[1] + M[a][2] * u[2];
    u[0] = v[0];
    u[1] = v[1];
    u[2] = v[2];
}

// readUnitCell in NCELL mode, fileio_fftw3.cpp:1313-1657, with replicateUnitCell :1188-1306
int boxed_super_cell(const std::vector<QAtom>& uc, const float* cube, float ctx, float cty, float ctz, float xOff, float yOff,
                     Tds& tds, std::vector<QAtom>& atoms, Cell& cell);

int build_super_cell(const char* file, int ncx, int ncy, int ncz, float ctx, float cty, float ctz, float xOff, float yOff,
                     const float* cube, bool tds_on, float tds_temp, std::vector<QAtom>& atoms, Cell& cell)
{
    // format by file-name ending, fileio_fftw3.cpp:1340-1368 (.pdb and .xyz are refused there too)
    const size_t n = std::strlen(file);
    auto ends = [&](const char* e) { return n >= std::strlen(e) && std::strcmp(file + n - std::strlen(e), e) == 0; };
    std::vector<QAtom> uc;
    int rc;
    if (ends(".cssr")) {
        rc = read_cssr_cell(file, cell);
        if (!rc) rc = read_cssr_atoms(file, cell.ncoord, uc);
    } else if (ends(".cfg")) {
        rc = read_cfg_cell(file, cell);
        if (!rc) rc = read_cfg_atoms(file, cell.ncoord, uc);
    } else if (ends(".dat")) {
        rc = read_dat_cell(file, cell);
        if (!rc) rc = read_dat_atoms(file, cell.ncoord, uc);
    } else
        return fail(FDES_EUNSUPPORTED, "cannot read anything else than .cssr, .cfg or .dat unit cells (no .pdb/.xyz)", file);
    if (rc) return rc;
    const int nc = cell.ncoord;
    if (ncx < 1 || ncy < 1 || ncz < 1 || (double)nc * ncx * ncy * ncz > 2.0e9) return fail(FDES_EINVAL, "bad NCELLX/Y/Z", nullptr);
    // atomCompareZYX, fileio_fftw3.cpp:109-123
    std::stable_sort(uc.begin(), uc.end(), [](const QAtom& a, const QAtom& b) {
        if (a.z != b.z) return a.z < b.z;
        if (a.y != b.y) return a.y < b.y;
        return a.x < b.x;
    });
    // boxed mode ("Cube:"): fileio_fftw3.cpp:1508-1513
    Tds tds;
    if (tds_on) tds.init(cell.Mm, tds_temp);
    if (cube[0] > 0 && cube[1] > 0 && cube[2] > 0) return boxed_super_cell(uc, cube, ctx, cty, ctz, xOff, yOff, tds, atoms, cell);
    // replicateUnitCell with handleVacancies, fileio_fftw3.cpp:1188-1306: sites are visited from the last sorted atom
    // backwards, atoms at one position (within 1e-6) form a site; a site with total occupancy below 1 or with several atoms
    // draws ONE number per cell - cells from the last to the first - and keeps the atom whose occupancy interval holds it
    // (none, if it falls beyond the sum), the others become vacancies (Znum 0, position and occupancy kept: they travel on
    // to the engine as species 0, as they do in the reference).  ran1 starts from its seed of -1, as in a fresh process.
    Ran1 rng;
    atoms.assign((size_t)nc * ncx * ncy * ncz, QAtom{});
    for (int i = nc - 1; i >= 0;) {
        int jequal = i - 1;
        double totOcc = 1;
        if (uc[(size_t)i].Znum > 0) {
            totOcc = uc[(size_t)i].occ;
            for (; jequal >= 0; jequal--) {
                if (std::fabs(uc[(size_t)i].x - uc[(size_t)jequal].x) < 1e-6 && std::fabs(uc[(size_t)i].y - uc[(size_t)jequal].y) < 1e-6 &&
                    std::fabs(uc[(size_t)i].z - uc[(size_t)jequal].z) < 1e-6)
                    totOcc += uc[(size_t)jequal].occ;
                else break;
            }
        }
        for (int icx = ncx - 1; icx >= 0; icx--)
            for (int icy = ncy - 1; icy >= 0; icy--)
                for (int icz = ncz - 1; icz >= 0; icz--) {
                    const size_t jCell = (size_t)(icz + icy * ncz + icx * ncy * ncz) * (size_t)nc;
                    for (int i2 = i; i2 > jequal; i2--) atoms[jCell + (size_t)i2] = uc[(size_t)i2];
                    int jChoice = i;
                    if (totOcc < 1 || jequal < i - 1) {
                        const double choice = totOcc < 1.0 ? rng.next() : totOcc * rng.next();
                        double lastOcc = 0;
                        for (int i2 = i; i2 > jequal; i2--) {
                            if (choice < lastOcc || choice >= lastOcc + uc[(size_t)i2].occ) atoms[jCell + (size_t)i2].Znum = 0; // vacancy
                            else jChoice = i2;
                            lastOcc += uc[(size_t)i2].occ;
                        }
                    }
                    double u[3];
                    tds.draw(uc[(size_t)jChoice].dw, u); // zero unless `tds: yes`; one displacement per site and cell
                    for (int i2 = i; i2 > jequal; i2--) {
                        QAtom& a = atoms[jCell + (size_t)i2];
                        a.x = (float)((double)(uc[(size_t)i2].x + (float)icx) + u[0]); // float + int, + double
                        a.y = (float)((double)(uc[(size_t)i2].y + (float)icy) + u[1]);
                        a.z = (float)((double)(uc[(size_t)i2].z + (float)icz) + u[2]);
                    }
                }
        i = jequal;
    }
    const double(*Mm)[3] = cell.Mm;
    for (QAtom& a : atoms) { // fractional -> cartesian with the transposed cell matrix
        const double x = Mm[0][0] * a.x + Mm[1][0] * a.y + Mm[2][0] * a.z;
        const double y = Mm[0][1] * a.x + Mm[1][1] * a.y + Mm[2][1] * a.z;
        const double z = Mm[0][2] * a.x + Mm[1][2] * a.y + Mm[2][2] * a.z;
        a.x = (float)x;
        a.y = (float)y;
        a.z = (float)z;
    }
    const double bc[3] = {ncx / 2.0, ncy / 2.0, ncz / 2.0};
    double ctr[3];
    for (int a = 0; a < 3; a++) ctr[a] = Mm[0][a] * bc[0] + Mm[1][a] * bc[1] + Mm[2][a] * bc[2];
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    bool first = true;
    for (int icx = 0; icx <= ncx; icx += ncx)
        for (int icy = 0; icy <= ncy; icy += ncy)
            for (int icz = 0; icz <= ncz; icz += ncz) {
                double u[3];
                for (int a = 0; a < 3; a++) u[a] = Mm[0][a] * (icx - bc[0]) + Mm[1][a] * (icy - bc[1]) + Mm[2][a] * (icz - bc[2]);
                rotate(u, ctx, cty, ctz);
                for (int a = 0; a < 3; a++) {
                    const double v = u[a] + ctr[a];
                    if (first) lo[a] = hi[a] = v;
                    else {
                        lo[a] = lo[a] > v ? v : lo[a];
                        hi[a] = hi[a] < v ? v : hi[a];
                    }
                }
                first = false;
            }
    if (ctx != 0 || cty != 0 || ctz != 0)
        for (QAtom& a : atoms) {
            double u[3] = {a.x - ctr[0], a.y - ctr[1], a.z - ctr[2]};
            rotate(u, ctx, cty, ctz);
            a.x = (float)(u[0] + ctr[0]);
            a.y = (float)(u[1] + ctr[1]);
            a.z = (float)(u[2] + ctr[2]);
        }
    for (QAtom& a : atoms) {
        a.x = (float)(a.x - lo[0]);
        a.y = (float)(a.y - lo[1]);
        a.z = (float)(a.z - lo[2]);
    }
    cell.ax = (float)(hi[0] - lo[0]);
    cell.by = (float)(hi[1] - lo[1]);
    cell.c = (float)(hi[2] - lo[2]);
    if (xOff != 0 || yOff != 0)
        for (QAtom& a : atoms) {
            a.x += xOff;
            a.y += yOff;
        }
    return FDES_OK;
}

// tiltBoxed, fileio_fftw3.cpp:1661-1925 (Einstein mode, no thermal displacements): the cell vectors are the COLUMNS of M,
// rotated by the crystal tilt (rotateMatrix: M <- R M, matrixlib.cpp:637-675); the lattice indices needed to reach every
// corner of the box [0, cube] - offset follow from inv(M) (inverse_3x3 with its identity fall-back for |det| < 0.0005,
// matrixlib.cpp:225-257); every site of every such cell is kept when its cartesian position + offset lies in the box
// (borders included), in the order site, ix, iy, iz.  A site with partial or shared occupancy draws one ran1 deviate per
// cell; the atom whose occupancy interval holds it lends its species / Debye-Waller factor / occupancy, and when none does
// the site's first atom is used (as there: jChoice starts as iatom and vacancies are only counted).
int boxed_super_cell(const std::vector<QAtom>& uc, const float* cube, float ctx, float cty, float ctz, float xOff, float yOff,
                     Tds& tds, std::vector<QAtom>& atoms, Cell& cell)
{
    const int nc = (int)uc.size();
    double M[9], Minv[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) M[a * 3 + b] = cell.Mm[b][a];
    if (ctx != 0 || cty != 0 || ctz != 0) { // the reference's cached rotation starts as the identity
        const double px = ctx, py = cty, pz = ctz;
        const double R[9] = {std::cos(pz) * std::cos(py), std::cos(pz) * std::sin(py) * std::sin(px) - std::sin(pz) * std::cos(px),
                             std::cos(pz) * std::sin(py) * std::cos(px) + std::sin(pz) * std::sin(px),
                             std::sin(pz) * std::cos(py), std::sin(pz) * std::sin(py) * std::sin(px) + std::cos(pz) * std::cos(px),
                             std::sin(pz) * std::sin(py) * std::cos(px) - std::cos(pz) * std::sin(px),
                             -std::sin(py), std::cos(py) * std::sin(px), std::cos(py) * std::cos(px)};
        double T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++) T[i * 3 + j] += R[i * 3 + k] * M[k * 3 + j];
        std::memcpy(M, T, sizeof(M));
    }
    {
        const double* a = M;
        double det = a[0] * (a[4] * a[8] - a[7] * a[5]) - a[1] * (a[3] * a[8] - a[6] * a[5]) + a[2] * (a[3] * a[7] - a[6] * a[4]);
        if (std::fabs(det) < 0.0005f) {
            const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            std::memcpy(Minv, I, sizeof(Minv));
        } else {
            det = 1.0f / det;
            Minv[0] = (a[4] * a[8] - a[5] * a[7]) * det;  Minv[1] = -(a[1] * a[8] - a[7] * a[2]) * det; Minv[2] = (a[1] * a[5] - a[4] * a[2]) * det;
            Minv[3] = -(a[3] * a[8] - a[5] * a[6]) * det; Minv[4] = (a[0] * a[8] - a[6] * a[2]) * det;  Minv[5] = -(a[0] * a[5] - a[3] * a[2]) * det;
            Minv[6] = (a[3] * a[7] - a[6] * a[4]) * det;  Minv[7] = -(a[0] * a[7] - a[6] * a[1]) * det; Minv[8] = (a[0] * a[4] - a[1] * a[3]) * det;
        }
    }
    auto mul = [](const double* m, const double* v, double* o) { // matrixProduct(m, 3, 3, v, 3, 1, o): sums start from 0.0
        for (int i = 0; i < 3; i++) {
            o[i] = 0.0;
            for (int k = 0; k < 3; k++) o[i] += m[i * 3 + k] * v[k];
        }
    };
    const double dx = xOff, dy = yOff, dz = 0;
    double a[3] = {0, 0, 0}, b[3];
    mul(Minv, a, b);
    int nxmin, nxmax, nymin, nymax, nzmin, nzmax;
    nxmin = nxmax = (int)std::floor(b[0] - dx);
    nymin = nymax = (int)std::floor(b[1] - dy);
    nzmin = nzmax = (int)std::floor(b[2] - dz);
    for (int ix = 0; ix <= 1; ix++)
        for (int iy = 0; iy <= 1; iy++)
            for (int iz = 0; iz <= 1; iz++) {
                a[0] = ix * cube[0] - dx; a[1] = iy * cube[1] - dy; a[2] = iz * cube[2] - dz;
                mul(Minv, a, b);
                if (nxmin > (int)std::floor(b[0])) nxmin = (int)std::floor(b[0]);
                if (nxmax < (int)std::ceil(b[0])) nxmax = (int)std::ceil(b[0]);
                if (nymin > (int)std::floor(b[1])) nymin = (int)std::floor(b[1]);
                if (nymax < (int)std::ceil(b[1])) nymax = (int)std::ceil(b[1]);
                if (nzmin > (int)std::floor(b[2])) nzmin = (int)std::floor(b[2]);
                if (nzmax < (int)std::ceil(b[2])) nzmax = (int)std::ceil(b[2]);
            }
    if ((double)(nxmax - nxmin + 1) * (nymax - nymin + 1) * (nzmax - nzmin + 1) * nc > 2.0e9) return fail(FDES_EINVAL, "'Cube:' box needs too many unit cells", nullptr);
    Ran1 rng;
    atoms.clear();
    for (int iatom = 0; iatom < nc;) {
        const QAtom& site = uc[(size_t)iatom];
        int jequal = iatom + 1;
        double totOcc = 1;
        if (site.Znum > 0) {
            totOcc = site.occ;
            for (; jequal < nc; jequal++) {
                if (std::fabs(site.x - uc[(size_t)jequal].x) < 1e-6 && std::fabs(site.y - uc[(size_t)jequal].y) < 1e-6 &&
                    std::fabs(site.z - uc[(size_t)jequal].z) < 1e-6)
                    totOcc += uc[(size_t)jequal].occ;
                else break;
            }
        }
        for (int ix = nxmin; ix <= nxmax; ix++)
            for (int iy = nymin; iy <= nymax; iy++)
                for (int iz = nzmin; iz <= nzmax; iz++) {
                    const double aO[3] = {(double)((float)ix + site.x), (double)((float)iy + site.y), (double)((float)iz + site.z)}; // int + float
                    int jChoice = iatom;
                    if (totOcc < 1 || jequal > iatom + 1) {
                        const double choice = totOcc < 1.0 ? rng.next() : totOcc * rng.next();
                        double lastOcc = 0;
                        for (int i2 = iatom; i2 < jequal; i2++) {
                            if (!(choice < lastOcc || choice >= lastOcc + uc[(size_t)i2].occ)) jChoice = i2;
                            lastOcc += uc[(size_t)i2].occ;
                        }
                    }
                    double u[3];
                    tds.draw(uc[(size_t)jChoice].dw, u); // (drawn for every cell, kept or not, as there)
                    mul(M, aO, b);
                    const double x = b[0] + dx, y = b[1] + dy, z = b[2] + dz;
                    if (x >= 0 && x <= cube[0] && y >= 0 && y <= cube[1] && z >= 0 && z <= cube[2]) { // the UNdisplaced position decides
                        const double ad[3] = {aO[0] + u[0], aO[1] + u[1], aO[2] + u[2]};
                        mul(M, ad, b);
                        QAtom n = uc[(size_t)jChoice];
                        n.x = (float)(b[0] + dx);
                        n.y = (float)(b[1] + dy);
                        n.z = (float)(b[2] + dz);
                        atoms.push_back(n);
                    }
                }
        iatom = jequal;
    }
    cell.ax = cube[0];
    cell.by = cube[1];
    cell.c = cube[2];
    return FDES_OK;
}

// wavelength(kev) in Angstroem, src/rwQsc.cu:1236-1247
double qstem_wavelength(double kev)
{
    const double emass = 510.99906, hc = 12.3984244;
    return hc / std::sqrt(kev * (2 * emass + kev));
}

bool yes(const std::string& s)
{
    char w[256] = "";
    std::sscanf(s.c_str(), "%255s", w);
    return std::tolower((unsigned char)w[0]) == 'y';
}

// "%g %s" with an optional unit: a leading 'd' (deg) converts to rad, src/rwQsc.cu:98-138
float angle(const std::string& s)
{
    const double pi = 3.1415926535897;
    float v = 0.f;
    char unit[256] = "";
    std::sscanf(s.c_str(), "%g %255s", &v, unit);
    if (std::tolower((unsigned char)unit[0]) == 'd') v = (float)(v * (pi / 180.0));
    return v;
}

bool file_exists(const std::string& p)
{
    FILE* f = std::fopen(p.c_str(), "r");
    if (f) std::fclose(f);
    return f != nullptr;
}

} // namespace

extern "C" int fdes_read_qsc(const char* file, fdes_params* p, fdes_atoms* atoms, int flags)
{
    if (!file || !p || !p->tiltbeam || p->cap < 1) return FDES_EINVAL;
    if (!(flags & FDES_CNF_SKIP_ATOMS) && !atoms) return FDES_EINVAL;
    ParFile q;
    if (!q.open(file)) return fail(FDES_EIO, "could not open input file", file);
    const double pi = 3.1415926535897;
    std::string r;

    if (q.find("mode:", r)) {
        if (r.find("TEM") == std::string::npos) return fail(FDES_EUNSUPPORTED, "FDES supports only TEM mode", file);
    } else
        return fail(FDES_EINVAL, "no 'mode:' (QSTEM then assumes STEM and needs scan parameters)", file);

    if (!q.find("filename:", r)) return fail(FDES_EINVAL, "no 'filename:' naming the crystal .cfg file", file);
    char word[kParBuf] = "";
    std::sscanf(r.c_str(), "%1023s", word);
    std::string cellFile = word;
    if (!cellFile.empty() && cellFile[0] == '"') { // quoted name, may contain blanks
        size_t a = r.find('"'), b = r.find('"', a + 1);
        cellFile = b == std::string::npos ? r.substr(a + 1) : r.substr(a + 1, b - a - 1);
    }
    int ncx = 1, ncy = 1, ncz = 1, cellDiv = 1;
    if (q.find("NCELLX:", r)) std::sscanf(r.c_str(), "%d", &ncx);
    if (q.find("NCELLY:", r)) std::sscanf(r.c_str(), "%d", &ncy);
    if (q.find("NCELLZ:", r)) { // "n" or "n/div"
        char a[256] = "";
        std::sscanf(r.c_str(), "%255s", a);
        if (char* s = std::strchr(a, '/')) {
            *s = '\0';
            cellDiv = std::atoi(s + 1);
        }
        ncz = std::atoi(a);
    }
    float btx = 0.f, bty = 0.f, ctx = 0.f, cty = 0.f, ctz = 0.f;
    if (q.find("Beam tilt X:", r)) btx = angle(r);
    if (q.find("Beam tilt Y:", r)) bty = angle(r);
    if (q.find("Crystal tilt X:", r)) ctx = angle(r);
    if (q.find("Crystal tilt Y:", r)) cty = angle(r);
    if (q.find("Crystal tilt Z:", r)) ctz = angle(r);
    float cube[3] = {0.f, 0.f, 0.f};
    if (q.find("Cube:", r)) std::sscanf(r.c_str(), "%g %g %g", &cube[0], &cube[1], &cube[2]);
    const bool boxed = cube[0] > 0 && cube[1] > 0 && cube[2] > 0;
    const bool tds_on = q.find("tds:", r) && yes(r);
    float tds_temp = 300.0f;
    if (q.find("temperature:", r)) std::sscanf(r.c_str(), "%g", &tds_temp);

    // atomPosFile: as given; without an extension ".cssr" is tried first, then ".cfg" (rwQsc.cu:181-210).  The
    // reference resolves it against the working directory; the .qsc's own directory is tried next.
    auto resolve = [&](const std::string& name) {
        std::string path = name;
        if (!file_exists(path)) {
            std::string dir = file;
            size_t slash = dir.find_last_of('/');
            if (slash != std::string::npos && name[0] != '/') path = dir.substr(0, slash + 1) + name;
        }
        return path;
    };
    if (cellFile.find('.') == std::string::npos) cellFile += file_exists(resolve(cellFile + ".cssr")) ? ".cssr" : ".cfg";
    std::string cellPath = resolve(cellFile);
    float xOff = 0.f, yOff = 0.f;
    if (q.find("xOffset:", r)) std::sscanf(r.c_str(), "%g", &xOff);
    if (q.find("yOffset:", r)) std::sscanf(r.c_str(), "%g", &yOff);

    std::vector<QAtom> sc;
    Cell cell;
    int rc = build_super_cell(cellPath.c_str(), ncx, ncy, ncz, ctx, cty, ctz, xOff, yOff, cube, tds_on, tds_temp, sc, cell);
    if (rc) return rc;

    int nx = 0, ny = 0;
    if (!q.find("nx:", r)) return fail(FDES_EINVAL, "no 'nx:'", file);
    std::sscanf(r.c_str(), "%d", &nx);
    if (q.find("ny:", r)) std::sscanf(r.c_str(), "%d", &ny);
    else ny = nx;
    float resX = 0.f, resY = 0.f, v0 = 0.f;
    if (q.find("resolutionX:", r)) std::sscanf(r.c_str(), "%g", &resX);
    if (q.find("resolutionY:", r)) std::sscanf(r.c_str(), "%g", &resY);
    if (!q.find("v0:", r)) return fail(FDES_EINVAL, "no 'v0:'", file);
    std::sscanf(r.c_str(), "%g", &v0);
    int centerSlices = 0;
    if (q.find("center slices:", r)) centerSlices = yes(r);
    // slice thickness / number of slices, src/rwQsc.cu:270-318 (boxed mode divides the box height instead of the cell's)
    float sliceTh = 0.f;
    int slices = 0;
    if (q.find("slice-thickness:", r)) {
        std::sscanf(r.c_str(), "%g", &sliceTh);
        if (q.find("slices:", r)) std::sscanf(r.c_str(), "%d", &slices);
        else slices = (int)((boxed ? cube[2] : cell.c) / (cellDiv * sliceTh) + 0.99);
        slices += centerSlices;
    } else if (q.find("slices:", r)) {
        std::sscanf(r.c_str(), "%d", &slices);
        if (slices == 1 && cellDiv == 1) sliceTh = boxed ? (float)(centerSlices ? 2.0 * cube[2] / cellDiv : cube[2] / cellDiv) : cell.c / cellDiv;
        else if (boxed) sliceTh = cube[2] / (cellDiv * slices - centerSlices); // (a zero divisor gives inf there as well)
        else if (slices > 0) sliceTh = cell.c / (cellDiv * slices);
    }
    if (slices <= 0) return fail(FDES_EINVAL, "number of slices = 0", file);
    if (nx < 1 || ny < 1) return fail(FDES_EINVAL, "bad nx/ny", file);
    if (resX == 0.0f) resX = (float)(cell.ax / (double)nx);
    if (resY == 0.0f) resY = (float)(cell.by / (double)ny);

    // probe / lens block, src/rwQsc.cu:521-620
    float Cs = 0.f, C5 = 0.f, df0 = 0.f, astigMag = 0.f, astigAngle = 0.f, alpha = 0.f;
    if (!q.find("Cs:", r)) return fail(FDES_EINVAL, "no 'Cs:'", file);
    std::sscanf(r.c_str(), "%g", &Cs);
    Cs = (float)(Cs * 1.0e7); // mm -> A
    if (q.find("C5:", r)) {
        std::sscanf(r.c_str(), "%g", &C5);
        C5 = (float)(C5 * 1.0e7);
    }
    df0 = -(float)std::sqrt(1.5 * Cs * qstem_wavelength(v0)); // Scherzer unless given
    if (q.find("defocus:", r)) {
        char a[256] = "";
        std::sscanf(r.c_str(), "%255s", a);
        const int c0 = std::tolower((unsigned char)a[0]);
        if (c0 == 's') df0 = -(float)std::sqrt(1.5 * Cs * qstem_wavelength(v0));
        else if (c0 == 'o') df0 = -(float)std::sqrt(Cs * qstem_wavelength(v0));
        else {
            std::sscanf(r.c_str(), "%g", &df0); // nm
            df0 = (float)(10.0 * df0);          // -> A
        }
    }
    if (q.find("astigmatism:", r)) std::sscanf(r.c_str(), "%g", &astigMag);
    astigMag = (float)(10.0 * astigMag);
    if (q.find("astigmatism angle:", r)) std::sscanf(r.c_str(), "%g", &astigAngle);
    astigAngle = (float)(astigAngle * (pi / 180.0));
    if (!q.find("alpha:", r)) return fail(FDES_EINVAL, "no 'alpha:'", file);
    std::sscanf(r.c_str(), "%g", &alpha);

    // mapping to params_t, src/rwQsc.cu:937-1012; `p` holds defaultParams (fdes_params_init) for the rest
    p->n3 = 1;
    p->n1 = nx;
    p->n2 = ny;
    p->dn1 = nx / 2; // round(float(nx/2)): the division is already integral
    p->dn2 = ny / 2;
    p->m1 = p->n1 + 2 * p->dn1;
    p->m2 = p->n2 + 2 * p->dn2;
    p->m3 = slices;
    p->d1 = (float)(resX * 1e-10);
    p->d2 = (float)(resY * 1e-10);
    p->d3 = (float)(sliceTh * 1e-10);
    p->subSlTh = (float)(sliceTh * 1e-10 / 10);
    p->tilt_offset_x = ctx;
    p->tilt_offset_y = cty;
    p->tilt_offset_z = ctz;
    p->tiltbeam[0] = btx;
    p->tiltbeam[1] = bty;
    p->E0 = (float)(v0 * 1e3);
    p->illangle = (float)(alpha / 1e3);
    p->ab.A1_0 = (float)(astigMag * 1e-9);
    p->ab.A1_1 = (float)(astigAngle * 1e-9);
    p->ab.C1_0 = (float)(df0 * 1e-10);
    p->ab.C3_0 = (float)(Cs * 1e-10);
    p->ab.C5_0 = (float)(C5 * 1e-3);
    {
        // rwQsc.cu:973-992 cuts the names at ".cfg" and ends the program (EXIT_FAILURE, :981) when the name holds
        // none, so a .cssr/.dat cell never gets past readQsc there; here those names are cut at their own ending.
        std::string mat = cellFile, name = cellFile;
        size_t dot = mat.find(".cfg");
        if (dot == std::string::npos) dot = mat.rfind(".cssr");
        if (dot == std::string::npos) dot = mat.rfind(".dat");
        if (dot == std::string::npos) return fail(FDES_EINVAL, "crystal file is not a .cfg, .cssr or .dat", cellFile.c_str());
        mat.resize(dot);
        char cellnum[64];
        std::snprintf(cellnum, sizeof(cellnum), "_CELL_%02d_%02d_%02d", ncx, ncy, ncz);
        name = name.substr(0, dot) + cellnum;
        std::snprintf(p->material, FDES_STR, "%s", mat.c_str());
        std::snprintf(p->sample_name, FDES_STR, "%s", name.c_str());
    }
    if (q.find("cal_mode:", r)) std::sscanf(r.c_str(), "%d", &p->mode);
    if (q.find("focus_spread:", r)) std::sscanf(r.c_str(), "%g", &p->defocspread);
    if (q.find("objective_aperture:", r)) std::sscanf(r.c_str(), "%g", &p->ObjAp);
    if (q.find("pixel_dose:", r)) std::sscanf(r.c_str(), "%g", &p->pD);
    if (q.find("absorptive_potential_factor:", r)) std::sscanf(r.c_str(), "%g", &p->imPot);
    if (q.find("mtf_a:", r)) std::sscanf(r.c_str(), "%g", &p->mtfa);
    if (q.find("mtf_b:", r)) std::sscanf(r.c_str(), "%g", &p->mtfb);
    if (q.find("mtf_c:", r)) std::sscanf(r.c_str(), "%g", &p->mtfc);
    if (q.find("frozen_phonons:", r)) std::sscanf(r.c_str(), "%d", &p->frPh);

    if (!(flags & FDES_CNF_SKIP_ATOMS)) { // atomsFromExternal == 0, src/rwQsc.cu:1014-1090
        const int nAt = (int)sc.size();
        rc = fdes_atoms_alloc(atoms, nAt);
        if (rc) return rc;
        p->nAt = nAt;
        float lo[3] = {1.f, 1.f, 1.f}, hi[3] = {0.f, 0.f, 0.f};
        for (int i = 0; i < nAt; i++) {
            const QAtom& a = sc[(size_t)i];
            atoms->Z[i] = a.Znum;
            atoms->dwf[i] = (float)(a.dw * 1e-20);
            atoms->occ[i] = a.occ;
            const float v[3] = {(float)(a.x * 1e-10), (float)(a.y * 1e-10), (float)(a.z * 1e-10)};
            for (int c = 0; c < 3; c++) {
                atoms->xyz[3 * i + c] = v[c];
                if (v[c] > hi[c]) hi[c] = v[c];
                if (v[c] < lo[c]) lo[c] = v[c];
            }
        }
        for (int i = 0; i < nAt; i++)
            for (int c = 0; c < 3; c++) atoms->xyz[3 * i + c] = atoms->xyz[3 * i + c] - (hi[c] - lo[c]) / 2;
    }
    return FDES_OK;
}
