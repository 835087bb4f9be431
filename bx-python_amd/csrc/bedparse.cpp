// bedparse.cpp -- BED text -> SoA columns on the host (the step before the hot path).
//
// The reference parses every line in Python: `line.split()` + `int()` per field
// (lib/bx/bitset_builders.py:33-46, scripts/bed_intersect.py:46-50); once the GPU answers a
// chromosome's queries in microseconds that loop is all that is left of the run time.
// This is a strict, single-pass C++ parser for the common case.  It never guesses: the moment a
// line is not plain ASCII BED -- a field that is not [+-]?[0-9]+, a '\r', a NUL, a byte >= 0x80, too few
// columns -- it STOPS and reports the line number, and the Python code takes over from that line
// with the reference's exact semantics (and exceptions).  Lines starting with '#' and
// whitespace-only lines are skipped, fields are split on runs of ASCII whitespace like
// str.split(); both as the reference does.
//
// No GPU code here: plain C++ compiled into libbxmi.so.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unistd.h>
#include <unordered_map>
#include <vector>

#include "common.hpp"

struct bxmi_bed {
    std::vector<int32_t> chrom;      // chromosome id per row (ids in first-appearance order)
    std::vector<int64_t> start, end;
    std::vector<int64_t> line_off;   // byte offset of the row's line in the parsed buffer
    std::vector<int32_t> line_len;   // length of the line including its '\n' (if any)
    std::vector<std::string> names;  // id -> chromosome name
    int64_t stop_line = -1;          // 0-based index (in lines of the buffer) of the first line NOT consumed, -1 = all consumed
    int64_t stop_off = -1;           // its byte offset
    int64_t lines_seen = 0;          // lines consumed, including skipped comment/blank lines
};

namespace {

inline bool py_space(unsigned char c)
{
    // bytes that str.split()/str.isspace() treat as whitespace in ASCII text
    return c == ' ' || (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x1f);
}

// [+-]?[0-9]+ fitting int64; anything else (underscores, unicode digits, empty) -> false
inline bool parse_i64(const char *p, const char *e, int64_t *out)
{
    if (p == e) return false;
    bool neg = false;
    if (*p == '+' || *p == '-') {
        neg = *p == '-';
        ++p;
        if (p == e) return false;
    }
    if (e - p > 18) return false;  // keep well inside int64; longer literals go to Python
    int64_t v = 0;
    for (; p < e; ++p) {
        unsigned d = (unsigned char)*p - '0';
        if (d > 9) return false;
        v = v * 10 + (int64_t)d;
    }
    *out = neg ? -v : v;
    return true;
}

}  // namespace

extern "C" int bxmi_bed_parse(const char *data, int64_t len, int chrom_col, int start_col, int end_col, bxmi_bed_t **out)
{
    if (!out || (len > 0 && !data) || len < 0 || chrom_col < 0 || start_col < 0 || end_col < 0)
        return bxmi::fail(BXMI_EINVAL, "bxmi_bed_parse: bad arguments");
    bxmi_bed *b = new (std::nothrow) bxmi_bed();
    if (!b) return bxmi::fail(BXMI_ENOMEM, "bxmi_bed_parse: host allocation failed");
    const int need = (chrom_col > start_col ? (chrom_col > end_col ? chrom_col : end_col) : (start_col > end_col ? start_col : end_col)) + 1;
    std::unordered_map<std::string, int32_t> ids;
    std::string last_name;
    int32_t last_id = -1;
    try {
        const char *p = data, *fin = data + len;
        size_t guess = (size_t)(len / 24) + 16;
        b->chrom.reserve(guess), b->start.reserve(guess), b->end.reserve(guess), b->line_off.reserve(guess), b->line_len.reserve(guess);
        while (p < fin) {
            const char *line = p;
            const char *nl = (const char *)memchr(p, '\n', (size_t)(fin - p));
            const char *eol = nl ? nl : fin;       // end of content
            const char *next = nl ? nl + 1 : fin;  // start of the next line
            bool plain = true, blank = true;
            for (const char *q = line; q < eol; ++q) {
                unsigned char c = (unsigned char)*q;
                if (c >= 0x80 || c == '\r' || c == 0) {  // (a NUL is an ordinary character to Python, but ends a C string: names would be cut)
                    plain = false;
                    break;
                }
                if (!py_space(c)) blank = false;
            }
            if (!plain) {
                b->stop_line = b->lines_seen, b->stop_off = line - data;
                break;
            }
            if (line < eol && *line == '#') {  // line.startswith("#")
                b->lines_seen++, p = next;
                continue;
            }
            if (blank) {
                // "\n" and "  \n" are isspace(); a final empty fragment never becomes a line at all
                b->lines_seen++, p = next;
                continue;
            }
            // split on whitespace runs
            const char *fs[3] = {nullptr, nullptr, nullptr}, *fe[3] = {nullptr, nullptr, nullptr};
            int col = 0;
            const char *q = line;
            while (q < eol && col < need) {
                while (q < eol && py_space((unsigned char)*q)) ++q;
                if (q >= eol) break;
                const char *s = q;
                while (q < eol && !py_space((unsigned char)*q)) ++q;
                if (col == chrom_col) fs[0] = s, fe[0] = q;
                if (col == start_col) fs[1] = s, fe[1] = q;
                if (col == end_col) fs[2] = s, fe[2] = q;
                ++col;
            }
            int64_t s64 = 0, e64 = 0;
            if (col < need || !parse_i64(fs[1], fe[1], &s64) || !parse_i64(fs[2], fe[2], &e64)) {
                b->stop_line = b->lines_seen, b->stop_off = line - data;  // let Python raise what the reference raises
                break;
            }
            size_t nlen = (size_t)(fe[0] - fs[0]);
            if (last_id < 0 || last_name.size() != nlen || memcmp(last_name.data(), fs[0], nlen) != 0) {
                last_name.assign(fs[0], nlen);
                auto it = ids.find(last_name);
                if (it == ids.end()) {
                    last_id = (int32_t)b->names.size();
                    ids.emplace(last_name, last_id);
                    b->names.push_back(last_name);
                } else {
                    last_id = it->second;
                }
            }
            b->chrom.push_back(last_id);
            b->start.push_back(s64);
            b->end.push_back(e64);
            b->line_off.push_back(line - data);
            b->line_len.push_back((int32_t)(next - line));
            b->lines_seen++;
            p = next;
        }
    } catch (const std::bad_alloc &) {
        delete b;
        return bxmi::fail(BXMI_ENOMEM, "bxmi_bed_parse: host allocation failed");
    }
    *out = b;
    return BXMI_OK;
}

// ---- table mode: what lib/bx/intervals/io.py's GenomicIntervalReader does per line -------------------------------
// (tabular/io.py:105-141 classifies the line, intervals/io.py:36-104 parses the row.)  Same contract as above -- consume
// what is certain, stop at the first line that is not -- with the reader's rules instead of the builders':
//   * a line is its text up to '\n'; an empty one is a blank (the reader yields Comment('') for it);
//   * a line starting with one of the comment prefixes ("#", "track ") is the header when it is the very first line
//     and a comment otherwise;
//   * anything else is a row: fields split on TAB only; the chromosome field must survive str.strip() unchanged, start
//     and end must be canonical integers (-?(0|[1-9][0-9]*), so that writing str(int(field)) back changes nothing), the
//     strand field -- when the row has one -- exactly "+" or "-", and start <= end.  A row that fails any of this is
//     left to the Python code with everything after it: it raises, fixes up or re-splits exactly as the reference does.
struct bxmi_tab {
    std::vector<uint8_t> kind;       // per LINE: 0 row, 1 blank, 2 comment, 3 header
    std::vector<int64_t> line_off;   // per line: byte offset and content length (without the newline)
    std::vector<int32_t> line_len;
    std::vector<int32_t> chrom;      // per line (rows only meaningful): chromosome id in first-appearance order
    std::vector<int64_t> start, end;
    std::vector<uint8_t> strand;     // '+', '-' or 0 = the row has no strand field
    std::vector<std::string> names;
    int64_t stop_off = -1;           // byte offset of the first line not consumed, -1 = everything consumed
};

extern "C" int bxmi_tab_parse(const char *data, int64_t len, int chrom_col, int start_col, int end_col, int strand_col,
                              const char *const *comment_prefixes, int n_prefixes, bxmi_tab_t **out)
{
    if (!out || (len > 0 && !data) || len < 0 || chrom_col < 0 || start_col < 0 || end_col < 0 || n_prefixes < 0 || (n_prefixes > 0 && !comment_prefixes))
        return bxmi::fail(BXMI_EINVAL, "bxmi_tab_parse: bad arguments");
    bxmi_tab *b = new (std::nothrow) bxmi_tab();
    if (!b) return bxmi::fail(BXMI_ENOMEM, "bxmi_tab_parse: host allocation failed");
    const int need = (chrom_col > start_col ? (chrom_col > end_col ? chrom_col : end_col) : (start_col > end_col ? start_col : end_col)) + 1;
    std::vector<std::pair<const char *, size_t>> pre;
    for (int i = 0; i < n_prefixes; i++) pre.emplace_back(comment_prefixes[i], strlen(comment_prefixes[i]));
    std::unordered_map<std::string, int32_t> ids;
    std::string last_name;
    int32_t last_id = -1;
    auto canonical_int = [](const char *p, const char *e, int64_t *v) {
        if (p == e) return false;
        const char *d = *p == '-' ? p + 1 : p;
        if (d == e || e - d > 18) return false;
        if (*d == '0' && (e - d > 1 || d != p)) return false;  // "007", "-0"
        int64_t x = 0;
        for (const char *q = d; q < e; ++q) {
            const unsigned c = (unsigned char)*q - '0';
            if (c > 9) return false;
            x = x * 10 + (int64_t)c;
        }
        *v = d != p ? -x : x;
        return true;
    };
    try {
        const char *p = data, *fin = data + len;
        const size_t guess = (size_t)(len / 24) + 16;
        b->kind.reserve(guess), b->line_off.reserve(guess), b->line_len.reserve(guess), b->chrom.reserve(guess), b->start.reserve(guess),
            b->end.reserve(guess), b->strand.reserve(guess);
        while (p < fin) {
            const char *line = p;
            const char *nl = (const char *)memchr(p, '\n', (size_t)(fin - p));
            const char *eol = nl ? nl : fin, *next = nl ? nl + 1 : fin;
            bool plain = true;
            for (const char *q = line; q < eol; ++q)
                if ((unsigned char)*q >= 0x80 || *q == '\r' || *q == 0) {
                    plain = false;
                    break;
                }
            uint8_t kind = 0;
            int32_t cid = -1;
            int64_t s64 = 0, e64 = 0;
            uint8_t strand = 0;
            if (plain && line == eol) {
                kind = 1;
            } else if (plain) {
                for (const auto &pr : pre)
                    if ((size_t)(eol - line) >= pr.second && memcmp(line, pr.first, pr.second) == 0) {
                        kind = b->kind.empty() ? 3 : 2;
                        break;
                    }
            }
            if (plain && kind == 0) {
                // the fields of interest among the TAB-separated ones
                const char *fs[4] = {nullptr, nullptr, nullptr, nullptr}, *fe[4] = {nullptr, nullptr, nullptr, nullptr};
                int col = 0;
                const char *s = line;
                for (const char *q = line;; ++q) {
                    if (q == eol || *q == '\t') {
                        if (col == chrom_col) fs[0] = s, fe[0] = q;
                        if (col == start_col) fs[1] = s, fe[1] = q;
                        if (col == end_col) fs[2] = s, fe[2] = q;
                        if (col == strand_col) fs[3] = s, fe[3] = q;
                        ++col;
                        s = q + 1;
                        if (q == eol) break;
                    }
                }
                plain = col >= need && canonical_int(fs[1], fe[1], &s64) && canonical_int(fs[2], fe[2], &e64) && s64 <= e64 &&
                        (fs[0] == fe[0] || (!py_space((unsigned char)*fs[0]) && !py_space((unsigned char)fe[0][-1])));
                if (plain && fs[3]) {
                    plain = fe[3] - fs[3] == 1 && (*fs[3] == '+' || *fs[3] == '-');
                    strand = plain ? (uint8_t)*fs[3] : 0;
                }
                if (plain) {
                    const size_t nlen = (size_t)(fe[0] - fs[0]);
                    if (last_id < 0 || last_name.size() != nlen || memcmp(last_name.data(), fs[0], nlen) != 0) {
                        last_name.assign(fs[0], nlen);
                        auto it = ids.find(last_name);
                        if (it == ids.end()) {
                            last_id = (int32_t)b->names.size();
                            ids.emplace(last_name, last_id);
                            b->names.push_back(last_name);
                        } else {
                            last_id = it->second;
                        }
                    }
                    cid = last_id;
                }
            }
            if (!plain) {
                b->stop_off = line - data;
                break;
            }
            b->kind.push_back(kind);
            b->line_off.push_back(line - data);
            b->line_len.push_back((int32_t)(eol - line));
            b->chrom.push_back(cid);
            b->start.push_back(s64);
            b->end.push_back(e64);
            b->strand.push_back(strand);
            p = next;
        }
    } catch (const std::bad_alloc &) {
        delete b;
        return bxmi::fail(BXMI_ENOMEM, "bxmi_tab_parse: host allocation failed");
    }
    *out = b;
    return BXMI_OK;
}

extern "C" int bxmi_tab_destroy(bxmi_tab_t *b)
{
    delete b;
    return BXMI_OK;
}

extern "C" int bxmi_tab_info(const bxmi_tab_t *b, int64_t *n_lines, int32_t *n_chroms, int64_t *stop_off)
{
    if (!b) return bxmi::fail(BXMI_EINVAL, "bxmi_tab_info: NULL handle");
    if (n_lines) *n_lines = (int64_t)b->kind.size();
    if (n_chroms) *n_chroms = (int32_t)b->names.size();
    if (stop_off) *stop_off = b->stop_off;
    return BXMI_OK;
}

extern "C" int bxmi_tab_columns(const bxmi_tab_t *b, const uint8_t **kind, const int64_t **line_off, const int32_t **line_len, const int32_t **chrom_id,
                                const int64_t **start, const int64_t **end, const uint8_t **strand)
{
    if (!b) return bxmi::fail(BXMI_EINVAL, "bxmi_tab_columns: NULL handle");
    if (kind) *kind = b->kind.data();
    if (line_off) *line_off = b->line_off.data();
    if (line_len) *line_len = b->line_len.data();
    if (chrom_id) *chrom_id = b->chrom.data();
    if (start) *start = b->start.data();
    if (end) *end = b->end.data();
    if (strand) *strand = b->strand.data();
    return BXMI_OK;
}

extern "C" const char *bxmi_tab_chrom_name(const bxmi_tab_t *b, int32_t id)
{
    if (!b || id < 0 || (size_t)id >= b->names.size()) return nullptr;
    return b->names[(size_t)id].c_str();
}

extern "C" int bxmi_bed_destroy(bxmi_bed_t *b)
{
    delete b;
    return BXMI_OK;
}

extern "C" int bxmi_bed_info(const bxmi_bed_t *b, int64_t *n_rows, int32_t *n_chroms, int64_t *stop_line, int64_t *stop_off,
                             int64_t *lines_seen)
{
    if (!b) return bxmi::fail(BXMI_EINVAL, "bxmi_bed_info: NULL handle");
    if (n_rows) *n_rows = (int64_t)b->start.size();
    if (n_chroms) *n_chroms = (int32_t)b->names.size();
    if (stop_line) *stop_line = b->stop_line;
    if (stop_off) *stop_off = b->stop_off;
    if (lines_seen) *lines_seen = b->lines_seen;
    return BXMI_OK;
}

extern "C" int bxmi_bed_columns(const bxmi_bed_t *b, const int32_t **chrom_id, const int64_t **start, const int64_t **end,
                                const int64_t **line_off, const int32_t **line_len)
{
    if (!b) return bxmi::fail(BXMI_EINVAL, "bxmi_bed_columns: NULL handle");
    if (chrom_id) *chrom_id = b->chrom.data();
    if (start) *start = b->start.data();
    if (end) *end = b->end.data();
    if (line_off) *line_off = b->line_off.data();
    if (line_len) *line_len = b->line_len.data();
    return BXMI_OK;
}

extern "C" const char *bxmi_bed_chrom_name(const bxmi_bed_t *b, int32_t id)
{
    if (!b || id < 0 || (size_t)id >= b->names.size()) return nullptr;
    return b->names[(size_t)id].c_str();
}

// Write the lines selected by mask[row] != 0, each followed by `suffix`, to file descriptor fd
// (bed_intersect.py:60,68 prints `line` + " ").  Buffered: one write(2) per 1 MiB.
extern "C" int bxmi_bed_emit_lines(const bxmi_bed_t *b, const char *data, const uint8_t *mask, const char *suffix, int fd)
{
    if (!b || !data || !mask) return bxmi::fail(BXMI_EINVAL, "bxmi_bed_emit_lines: bad arguments");
    FILE *f = fdopen(dup(fd), "w");
    if (!f) return bxmi::fail(BXMI_EINVAL, "bxmi_bed_emit_lines: cannot open fd %d: %s", fd, strerror(errno));
    setvbuf(f, nullptr, _IOFBF, 1 << 20);
    const size_t sl = suffix ? strlen(suffix) : 0;
    const size_t n = b->start.size();
    for (size_t i = 0; i < n; i++) {
        if (!mask[i]) continue;
        fwrite(data + b->line_off[i], 1, (size_t)b->line_len[i], f);
        if (sl) fwrite(suffix, 1, sl, f);
    }
    int rc = fclose(f);
    return rc == 0 ? BXMI_OK : bxmi::fail(BXMI_EINVAL, "bxmi_bed_emit_lines: write failed: %s", strerror(errno));
}
