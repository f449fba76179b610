// Host-side readers for on-disk matrices: Matrix Market (coordinate) and PETSc binary (MatView to a binary viewer).
// Matrix Market reader (the reference has no file I/O at all, so real matrices could not
// reach its solvers; this feeds CsrMatrix::from_csr / kryst_csr_create).  No GPU involved.
//
// Supported: "%%MatrixMarket matrix coordinate {real|integer|pattern} {general|symmetric|skew-symmetric}".
// Entries are sorted by (row, column); symmetric / skew-symmetric files are expanded to full storage; duplicate
// coordinates are summed in file order (assembled files have none).  pattern entries get the value 1.0.
#include "common.h"
#include <algorithm>
#include <cctype>
#include <exception>
#include <fstream>
#include <sstream>

namespace {
struct Entry { int64_t r, c; double v; int64_t seq; };

bool parse(const char* path, int64_t& nr, int64_t& nc, std::vector<Entry>& e) {
    std::ifstream f(path);
    if (!f) { kr::set_error("matrix market: cannot open %s", path); return false; }
    std::string line;
    if (!std::getline(f, line)) { kr::set_error("matrix market: empty file"); return false; }
    std::string lower(line);
    std::transform(lower.begin(), lower.end(), lower.begin(), [](unsigned char ch) { return (char)std::tolower(ch); });
    std::istringstream hs(lower);
    std::string banner, object, format, field, symmetry;
    hs >> banner >> object >> format >> field >> symmetry;
    if (banner != "%%matrixmarket" || object != "matrix" || format != "coordinate") {
        kr::set_error("matrix market: only 'matrix coordinate' files are supported"); return false;
    }
    const bool pattern = field == "pattern";
    if (!pattern && field != "real" && field != "integer") { kr::set_error("matrix market: field '%s' not supported", field.c_str()); return false; }
    const bool sym = symmetry == "symmetric", skew = symmetry == "skew-symmetric";
    if (!sym && !skew && symmetry != "general") { kr::set_error("matrix market: symmetry '%s' not supported", symmetry.c_str()); return false; }
    while (std::getline(f, line)) { if (!line.empty() && line[0] != '%') break; }
    int64_t nz = 0;
    { std::istringstream ss(line); if (!(ss >> nr >> nc >> nz) || nr < 0 || nc < 0 || nz < 0) { kr::set_error("matrix market: bad size line"); return false; } }
    e.clear(); e.reserve((size_t)(sym || skew ? 2 * nz : nz));
    for (int64_t k = 0; k < nz; ++k) {
        int64_t r, c; double v = 1.0;
        if (!(f >> r >> c)) { kr::set_error("matrix market: entry %lld missing", (long long)k); return false; }
        if (!pattern && !(f >> v)) { kr::set_error("matrix market: value of entry %lld missing", (long long)k); return false; }
        if (r < 1 || r > nr || c < 1 || c > nc) { kr::set_error("matrix market: entry %lld out of range", (long long)k); return false; }
        e.push_back(Entry{r - 1, c - 1, v, (int64_t)e.size()});
        if ((sym || skew) && r != c) e.push_back(Entry{c - 1, r - 1, skew ? -v : v, (int64_t)e.size()});
    }
    std::sort(e.begin(), e.end(), [](const Entry& a, const Entry& b) { return a.r != b.r ? a.r < b.r : (a.c != b.c ? a.c < b.c : a.seq < b.seq); });
    size_t w = 0;                                            // sum duplicates in file order
    for (size_t k = 0; k < e.size(); ++k) {
        if (w > 0 && e[w - 1].r == e[k].r && e[w - 1].c == e[k].c) e[w - 1].v = e[w - 1].v + e[k].v;
        else e[w++] = e[k];
    }
    e.resize(w);
    return true;
}
}  // namespace

static int64_t read_matrix_market_impl(const char* path, int64_t* nrows, int64_t* ncols, int64_t* row_ptr,
                                       int64_t* col_idx, double* vals) {
    if (!path || !nrows || !ncols) { kr::set_error("bad argument: read_matrix_market"); return -1; }
    std::vector<Entry> e;
    int64_t nr = 0, nc = 0;
    if (!parse(path, nr, nc, e)) return -1;
    *nrows = nr; *ncols = nc;
    if (row_ptr && col_idx && vals) {
        for (int64_t i = 0; i <= nr; ++i) row_ptr[i] = 0;
        for (const Entry& x : e) row_ptr[x.r + 1] += 1;
        for (int64_t i = 0; i < nr; ++i) row_ptr[i + 1] += row_ptr[i];
        for (size_t k = 0; k < e.size(); ++k) { col_idx[k] = e[k].c; vals[k] = e[k].v; }
    }
    return (int64_t)e.size();
}


// PETSc binary matrix (MatLoad / MatView with a binary viewer, AIJ): big-endian  int32 classid 1211216, rows, cols, nnz,
// int32 row lengths[rows], int32 column indices[nnz], float64 values[nnz].  Rows keep the file's entry order, which PETSc
// writes sorted by column; unsorted or duplicate columns are rejected by kryst_csr_create later, like any other input.
namespace {
bool read_be32(std::ifstream& f, int32_t* out, size_t count) {
    std::vector<unsigned char> buf(4 * count);
    if (!f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size())) return false;
    for (size_t k = 0; k < count; ++k)
        out[k] = (int32_t)(((uint32_t)buf[4 * k] << 24) | ((uint32_t)buf[4 * k + 1] << 16) | ((uint32_t)buf[4 * k + 2] << 8) | (uint32_t)buf[4 * k + 3]);
    return true;
}
bool read_be64f(std::ifstream& f, double* out, size_t count) {
    std::vector<unsigned char> buf(8 * count);
    if (!f.read(reinterpret_cast<char*>(buf.data()), (std::streamsize)buf.size())) return false;
    for (size_t k = 0; k < count; ++k) {
        uint64_t v = 0;
        for (int b = 0; b < 8; ++b) v = (v << 8) | buf[8 * k + b];
        memcpy(&out[k], &v, 8);
    }
    return true;
}
}  // namespace

static int64_t read_petsc_binary_impl(const char* path, int64_t* nrows, int64_t* ncols, int64_t* row_ptr,
                                      int64_t* col_idx, double* vals) {
    if (!path || !nrows || !ncols) { kr::set_error("bad argument: read_petsc_binary"); return -1; }
    std::ifstream f(path, std::ios::binary);
    if (!f) { kr::set_error("petsc binary: cannot open %s", path); return -1; }
    int32_t hdr[4];
    if (!read_be32(f, hdr, 4)) { kr::set_error("petsc binary: short header"); return -1; }
    if (hdr[0] != 1211216) { kr::set_error("petsc binary: class id %d is not a matrix (1211216)", hdr[0]); return -1; }
    if (hdr[1] < 0 || hdr[2] < 0 || hdr[3] < 0) { kr::set_error("petsc binary: negative sizes (a dense MATSEQDENSE file has nnz = -1)"); return -1; }
    const int64_t nr = hdr[1], nc = hdr[2], nz = hdr[3];
    *nrows = nr; *ncols = nc;
    if (!(row_ptr && col_idx && vals)) return nz;
    std::vector<int32_t> len((size_t)nr), col((size_t)nz);
    if (!read_be32(f, len.data(), (size_t)nr) || !read_be32(f, col.data(), (size_t)nz) || !read_be64f(f, vals, (size_t)nz)) {
        kr::set_error("petsc binary: file shorter than its header says"); return -1;
    }
    row_ptr[0] = 0;
    for (int64_t i = 0; i < nr; ++i) {
        if (len[(size_t)i] < 0) { kr::set_error("petsc binary: negative row length"); return -1; }
        row_ptr[i + 1] = row_ptr[i] + len[(size_t)i];
    }
    if (row_ptr[nr] != nz) { kr::set_error("petsc binary: row lengths do not add up to nnz"); return -1; }
    for (int64_t k = 0; k < nz; ++k) {
        if (col[(size_t)k] < 0 || col[(size_t)k] >= nc) { kr::set_error("petsc binary: column %d out of range", col[(size_t)k]); return -1; }
        col_idx[k] = col[(size_t)k];
    }
    return nz;
}


// Nothing unwinds across the C ABI: a file whose header promises more entries than memory holds (std::bad_alloc,
// std::length_error from the reserve) or any other C++ exception becomes -1 with a message.
template <class F>
static int64_t no_throw(const char* what, F f) {
    try { return f(); }
    catch (const std::exception& e) { kr::set_error("%s: %s", what, e.what()); return -1; }
    catch (...) { kr::set_error("%s: unknown C++ exception", what); return -1; }
}

extern "C" int64_t kryst_host_read_matrix_market(const char* path, int64_t* nrows, int64_t* ncols, int64_t* row_ptr,
                                                 int64_t* col_idx, double* vals) {
    return no_throw("matrix market reader", [&] { return read_matrix_market_impl(path, nrows, ncols, row_ptr, col_idx, vals); });
}

extern "C" int64_t kryst_host_read_petsc_binary(const char* path, int64_t* nrows, int64_t* ncols, int64_t* row_ptr,
                                                int64_t* col_idx, double* vals) {
    return no_throw("petsc binary reader", [&] { return read_petsc_binary_impl(path, nrows, ncols, row_ptr, col_idx, vals); });
}
