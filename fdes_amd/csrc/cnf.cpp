// cnf.cpp — reader / writer of FDES ".cnf" parameter files (host only).
//
// Input surface of readConfig / getParams / numberOfAtoms / readCoordinates
// (src/paramStructure.cu:42-302, 600-635, 1019-1077): one "key: value [value]" per line,
// matching on the first whitespace-delimited token; everything after the values is ignored
// (that is how '#' comments work); unknown keys are skipped silently.
//
// Two reading modes:
//   clean           whole lines of any length, blank lines ignored, every line seen once.
//   bug-compatible  (FDES_CNF_BUG_COMPATIBLE) the reference's control flow: 100-byte chunks for
//                   the parameter pass and 200-byte chunks for the atom passes, a token that
//                   stays "sticky" over blank lines and over the failed read at end of file.
//                   Consequences reproduced: a file ending in '\n' after the last "atom:" line
//                   yields the last atom twice; a blank line after a specimen_tilt:/beam_tilt:/
//                   defoci: line advances that index; a blank line after an "atom:" line adds
//                   one more atom (uninitialised memory in the reference; all-zero here).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "fdes_internal.h"

namespace {

struct KeyF { const char* key; int n; int nval; float fdes_params::*a; float fdes_aberration::*ab0; float fdes_aberration::*ab1; };

bool starts(const char* tok, const char* key, size_t n) { return std::strncmp(tok, key, n) == 0; }

// Apply one (token, line) pair to the parameter set. Index-advancing keys use idx[0..2].
void apply_line(const char* tok, const char* line, fdes_params* p, int* idx)
{
#define F1(KEY, N, DST) if (starts(tok, KEY, N)) { std::sscanf(line, "%*s %g", &(DST)); }
#define F2(KEY, N, D0, D1) if (starts(tok, KEY, N)) { std::sscanf(line, "%*s %g %g", &(D0), &(D1)); }
#define I1(KEY, N, DST) if (starts(tok, KEY, N)) { int v_; if (std::sscanf(line, "%*s %i", &v_) == 1) (DST) = v_; }
#define STR(KEY, N, FMT, DST) if (starts(tok, KEY, N)) { std::sscanf(line, FMT, (DST)); }
    F1("voltage:", 8, p->E0)
    F1("C1:", 3, p->ab.C1_0)
    F2("A1:", 3, p->ab.A1_0, p->ab.A1_1)
    F2("A2:", 3, p->ab.A2_0, p->ab.A2_1)
    F2("B2:", 3, p->ab.B2_0, p->ab.B2_1)
    F1("C3:", 3, p->ab.C3_0)
    F2("A3:", 3, p->ab.A3_0, p->ab.A3_1)
    F2("S3:", 3, p->ab.S3_0, p->ab.S3_1)
    F2("A4:", 3, p->ab.A4_0, p->ab.A4_1)
    F2("B4:", 3, p->ab.B4_0, p->ab.B4_1)
    F2("D4:", 3, p->ab.D4_0, p->ab.D4_1)
    F1("C5:", 3, p->ab.C5_0)
    F2("A5:", 3, p->ab.A5_0, p->ab.A5_1)
    F2("R5:", 3, p->ab.R5_0, p->ab.R5_1)
    F2("S5:", 3, p->ab.S5_0, p->ab.S5_1)
    F1("focus_spread:", 12, p->defocspread)
    F1("illumination_angle:", 19, p->illangle)
    F1("mtf_a:", 6, p->mtfa)
    F1("mtf_b:", 6, p->mtfb)
    F1("mtf_c:", 6, p->mtfc)
    F1("mtf_d:", 6, p->mtfd)
    F1("objective_aperture:", 19, p->ObjAp)
    I1("sample_size_x:", 14, p->m1)
    I1("sample_size_y:", 14, p->m2)
    I1("sample_size_z:", 14, p->m3)
    F1("pixel_size_x:", 13, p->d1)
    F1("pixel_size_y:", 13, p->d2)
    F1("pixel_size_z:", 13, p->d3)
    I1("border_size_x:", 14, p->dn1)
    I1("border_size_y:", 14, p->dn2)
    I1("image_size_x:", 13, p->n1)
    I1("image_size_y:", 13, p->n2)
    I1("image_size_z:", 13, p->n3)
    if (starts(tok, "specimen_tilt:", 14)) {
        if (idx[0] < p->cap) std::sscanf(line, "%*s %g %g", &p->tiltspec[2 * idx[0]], &p->tiltspec[2 * idx[0] + 1]);
        idx[0]++;
    }
    if (starts(tok, "beam_tilt:", 10)) {
        if (idx[1] < p->cap) std::sscanf(line, "%*s %g %g", &p->tiltbeam[2 * idx[1]], &p->tiltbeam[2 * idx[1] + 1]);
        idx[1]++;
    }
    if (starts(tok, "defoci:", 7)) {
        if (idx[2] < p->cap) std::sscanf(line, "%*s %g", &p->defoci[idx[2]]);
        idx[2]++;
    }
    STR("user_name:", 11, "user_name: %1023[^\n]", p->user_name)
    STR("institution:", 12, "institution: %1023[^\n]", p->institution)
    STR("department:", 11, "department: %1023[^\n]", p->department)
    STR("email:", 6, "email: %1023[^\n]", p->email)
    STR("comment:", 8, "comment: %1023[^\n]", p->comments)
    STR("sample_name:", 12, "sample_name: %1023[^\n]", p->sample_name)
    STR("material:", 9, "material: %1023[^\n]", p->material)
    F1("absorptive_potential_factor:", 28, p->imPot)
    F1("pixel_dose:", 11, p->pD)
    I1("frozen_phonons:", 15, p->frPh)
    F1("subpixel_size_z:", 16, p->subSlTh)
    F1("specimen_tilt_offset_x:", 23, p->tilt_offset_x)
    F1("specimen_tilt_offset_y:", 23, p->tilt_offset_y)
    F1("specimen_tilt_offset_z:", 23, p->tilt_offset_z)
    I1("mode:", 5, p->mode)
#undef F1
#undef F2
#undef I1
#undef STR
}

struct AtomRec { int Z; float x, y, z, dwf, occ; };

bool parse_atom(const char* line, AtomRec* r)
{
    AtomRec t = {0, 0, 0, 0, 0, 0};
    int n = std::sscanf(line, "%*s %i %g %g %g %g %g", &t.Z, &t.x, &t.y, &t.z, &t.dwf, &t.occ);
    *r = t;
    return n == 6;
}

// Chunked reader with the reference's "sticky token" semantics.
template <class Fn> void sticky_pass(FILE* fr, int chunk, Fn fn)
{
    std::vector<char> line(chunk, 0), tok(chunk + 1, 0);
    std::rewind(fr);
    std::clearerr(fr);
    do {
        if (std::fgets(line.data(), chunk, fr) != nullptr) std::sscanf(line.data(), "%s", tok.data());
        fn(tok.data(), line.data());
    } while (!std::feof(fr));
}

bool read_line(FILE* fr, std::string* out)
{
    out->clear();
    int c;
    bool any = false;
    while ((c = std::fgetc(fr)) != EOF) {
        any = true;
        if (c == '\n') break;
        out->push_back((char)c);
    }
    return any;
}

} // namespace

extern "C" int fdes_read_cnf(const char* file, fdes_params* p, fdes_atoms* atoms, int flags)
{
    if (!file || !p || !p->tiltspec || !p->tiltbeam || !p->defoci) return FDES_EINVAL;
    FILE* fr = std::fopen(file, "rt");
    if (!fr) return FDES_EIO;
    const bool compat = (flags & FDES_CNF_BUG_COMPATIBLE) != 0;
    int idx[3] = {0, 0, 0};
    std::vector<AtomRec> recs;
    if (compat) {
        sticky_pass(fr, 100, [&](const char* tok, const char* line) { apply_line(tok, line, p, idx); });
        if (atoms && !(flags & FDES_CNF_SKIP_ATOMS)) {
            // numberOfAtoms / readCoordinates see the same sequence of (token, line) pairs; the
            // second one blanks line[0] after each iteration, which does not change the parse.
            std::vector<char> line(200, 0), tok(201, 0);
            std::rewind(fr);
            std::clearerr(fr);
            while (!std::feof(fr)) {
                if (std::fgets(line.data(), 200, fr) != nullptr) std::sscanf(line.data(), "%s", tok.data());
                if (starts(tok.data(), "atom:", 5)) {
                    AtomRec r;
                    parse_atom(line.data(), &r); // failed parse -> all-zero record
                    recs.push_back(r);
                }
                line[0] = '#';
            }
        }
    } else {
        std::string ln;
        char tok[256];
        while (read_line(fr, &ln)) {
            tok[0] = 0;
            if (std::sscanf(ln.c_str(), "%255s", tok) != 1) continue; // blank line
            if (starts(tok, "atom:", 5)) {
                if (atoms && !(flags & FDES_CNF_SKIP_ATOMS)) {
                    AtomRec r;
                    if (parse_atom(ln.c_str(), &r)) recs.push_back(r);
                }
                continue;
            }
            apply_line(tok, ln.c_str(), p, idx);
        }
    }
    std::fclose(fr);
    if (p->n3 < 1 || p->n3 > p->cap) return FDES_EINVAL;
    if (atoms && !(flags & FDES_CNF_SKIP_ATOMS)) {
        int rc = fdes_atoms_alloc(atoms, (int)recs.size());
        if (rc) return rc;
        for (size_t i = 0; i < recs.size(); i++) {
            atoms->Z[i] = recs[i].Z;
            atoms->xyz[3 * i + 0] = recs[i].x;
            atoms->xyz[3 * i + 1] = recs[i].y;
            atoms->xyz[3 * i + 2] = recs[i].z;
            atoms->dwf[i] = recs[i].dwf;
            atoms->occ[i] = recs[i].occ;
        }
        p->nAt = (int)recs.size();
    }
    return FDES_OK;
}

// Parameter echo ("dataFDES_used.cnf", src/paramStructure.cu:362-491, 629-631).  Same keys in the
// same order and number format (%14.8g); atoms are written as re-readable "atom:" lines.
extern "C" int fdes_write_cnf(const char* file, const fdes_params* p, const fdes_atoms* a)
{
    if (!file || !p) return FDES_EINVAL;
    FILE* fw = std::fopen(file, "wt");
    if (!fw) return FDES_EIO;
    auto f1 = [&](const char* k, float v, const char* c) { std::fprintf(fw, "%s  %14.8g  %s\n", k, v, c); };
    auto f2 = [&](const char* k, float v, float w, const char* c) { std::fprintf(fw, "%s  %14.8g %14.8g  %s\n", k, v, w, c); };
    auto i1 = [&](const char* k, int v, const char* c) { std::fprintf(fw, "%s  %i  %s\n", k, v, c); };
    std::fprintf(fw, "# FDES parameter echo (MI355X engine). '#' starts a comment.\n\n# Constants\n");
    f1("m0:", 9.109389e-31f, "# kg");
    f1("c:", 299792458.0f, "# m/s");
    f1("e:", 1.602177e-19f, "# C");
    f1("h:", 6.626075e-34f, "# Js");
    f1("pi:", 3.141592654f, "");
    std::fprintf(fw, "\n# User\nuser_name: %s\ninstitution: %s\ndepartment: %s\nemail: %s\n\ncomment: %s\n", p->user_name,
                 p->institution, p->department, p->email, p->comments);
    std::fprintf(fw, "\n# Microscope\n");
    f1("voltage:", p->E0, "# V");
    f1("gamma:", p->gamma, "# derived");
    f1("lambda:", p->lambda, "# m, derived");
    f1("sigma:", p->sigma, "# 1/(Vm), derived");
    f1("focus_spread:", p->defocspread, "# m");
    f1("illumination_angle:", p->illangle, "# rad");
    f1("mtf_a:", p->mtfa, "");
    f1("mtf_b:", p->mtfb, "");
    f1("mtf_c:", p->mtfc, "");
    f1("mtf_d:", p->mtfd, "");
    f1("objective_aperture:", p->ObjAp, "# rad (radius)");
    std::fprintf(fw, "\n# Aberrations: amplitude [m], angle [rad]\n");
    const fdes_aberration& b = p->ab;
    f2("C1:", b.C1_0, b.C1_1, ""); f2("A1:", b.A1_0, b.A1_1, ""); f2("A2:", b.A2_0, b.A2_1, "");
    f2("B2:", b.B2_0, b.B2_1, ""); f2("C3:", b.C3_0, b.C3_1, ""); f2("A3:", b.A3_0, b.A3_1, "");
    f2("S3:", b.S3_0, b.S3_1, ""); f2("A4:", b.A4_0, b.A4_1, ""); f2("B4:", b.B4_0, b.B4_1, "");
    f2("D4:", b.D4_0, b.D4_1, ""); f2("C5:", b.C5_0, b.C5_1, ""); f2("A5:", b.A5_0, b.A5_1, "");
    f2("R5:", b.R5_0, b.R5_1, ""); f2("S5:", b.S5_0, b.S5_1, "");
    std::fprintf(fw, "\n# Imaging\n");
    i1("mode:", p->mode, "# 0 image, 1 diffraction, 2 CBED");
    i1("sample_size_x:", p->m1, "");
    i1("sample_size_y:", p->m2, "");
    i1("sample_size_z:", p->m3, "");
    f1("pixel_size_x:", p->d1, "# m");
    f1("pixel_size_y:", p->d2, "# m");
    f1("pixel_size_z:", p->d3, "# m");
    i1("border_size_x:", p->dn1, "");
    i1("border_size_y:", p->dn2, "");
    i1("image_size_x:", p->n1, "");
    i1("image_size_y:", p->n2, "");
    i1("image_size_z:", p->n3, "");
    f1("specimen_tilt_offset_x:", p->tilt_offset_x, "# rad");
    f1("specimen_tilt_offset_y:", p->tilt_offset_y, "# rad");
    f1("specimen_tilt_offset_z:", p->tilt_offset_z, "# rad");
    i1("frozen_phonons:", p->frPh, "");
    f1("pixel_dose:", p->pD, "# electrons / pixel");
    f1("subpixel_size_z:", p->subSlTh, "# m");
    std::fprintf(fw, "\n# Sample\nsample_name: %s\nmaterial: %s\n", p->sample_name, p->material);
    f1("absorptive_potential_factor:", p->imPot, "");
    std::fprintf(fw, "\n# Per-measurement specimen tilts, beam tilts [rad] and defoci [m]\n");
    for (int i = 0; i < p->n3 && i < p->cap; i++) f2("specimen_tilt:", p->tiltspec[2 * i], p->tiltspec[2 * i + 1], "");
    for (int i = 0; i < p->n3 && i < p->cap; i++) f2("beam_tilt:", p->tiltbeam[2 * i], p->tiltbeam[2 * i + 1], "");
    for (int i = 0; i < p->n3 && i < p->cap; i++) f1("defoci:", p->defoci[i], "");
    if (a) {
        std::fprintf(fw, "\n# Atoms: Z  x y z [m]  Debye-Waller [m^2]  occupancy\nNumber of atoms: %d\n", a->nAt);
        for (int j = 0; j < a->nAt; j++)
            std::fprintf(fw, "atom: %i %14.8g %14.8g %14.8g %14.8g %14.8g\n", a->Z[j], a->xyz[3 * j], a->xyz[3 * j + 1],
                         a->xyz[3 * j + 2], a->dwf[j], a->occ[j]);
    }
    std::fclose(fw);
    return FDES_OK;
}
