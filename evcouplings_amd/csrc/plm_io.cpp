// Host-side input plumbing of the couplings stage (no device code): the FASTA / A2M framing and the column encoding that
// evcouplings_amd/alignment_io.py needs before the alignment can go to the GPU.  plmc reads the alignment file itself
// (evcouplings/couplings/tools.py:202-262 only passes the path); here the Python host reads it, and at the headline
// (50 000 sequences x 300 columns, 16 MB of text) a per-line Python loop + fancy indexing was 0.23 s -- a fifth of a
// whole -g run.  These two functions are that work as two single passes (0.03 s).  Same rules as the Python
// restatement kept next to the caller (alignment_io.read_fasta_records, the fallback and the test oracle):
//   * every line is stripped of ASCII whitespace at both ends (space, \t, \n, \v, \f, \r); empty lines are skipped;
//   * a stripped line that starts with '>' opens a record, its id is the rest of the line;
//   * the stripped data lines of a record are concatenated (whitespace INSIDE a line stays);
//   * data before the first header is an error.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/plm_hip.h"
#include "plm_internal.h"

int plm_fail(int code, const char *fmt, ...);   // plm_host.cpp: records the message for plm_last_error()

namespace {
inline bool is_ws(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
}  // namespace

// Pass over the file image.  With seq_out == NULL only counts: *n_records, *seq_bytes (total stripped sequence bytes).
// With buffers: hdr_off / hdr_len = the id of record r (bytes of the header line behind '>', stripped), seq_len[r] = its
// stripped sequence length, seq_out = all sequences back to back.  Returns PLM_OK, or PLM_EINVAL for data before the
// first header.
extern "C" int plm_fasta_split(const char *buf, int64_t n, int64_t *n_records, int64_t *seq_bytes, int64_t *hdr_off,
                               int32_t *hdr_len, int64_t *seq_len, char *seq_out) {
    if (!buf || n < 0 || !n_records || !seq_bytes) return plm_fail(PLM_EINVAL, "plm_fasta_split: NULL argument");
    int64_t rec = 0, out = 0, pos = 0;
    while (pos < n) {
        const char *nl = (const char *)memchr(buf + pos, '\n', (size_t)(n - pos));
        const int64_t end = nl ? (nl - buf) : n;
        int64_t a = pos, b = end;
        while (a < b && is_ws((unsigned char)buf[a])) a++;
        while (b > a && is_ws((unsigned char)buf[b - 1])) b--;
        if (b > a) {
            if (buf[a] == '>') {
                if (seq_out) {
                    hdr_off[rec] = a + 1;
                    hdr_len[rec] = (int32_t)(b - a - 1);
                    seq_len[rec] = 0;
                }
                rec++;
            } else {
                if (rec == 0) return plm_fail(PLM_EINVAL, "sequence data before the first '>' header");
                if (seq_out) {
                    memcpy(seq_out + out, buf + a, (size_t)(b - a));
                    seq_len[rec - 1] += b - a;
                }
                out += b - a;
            }
        }
        pos = end + 1;
    }
    *n_records = rec;
    *seq_bytes = out;
    return PLM_OK;
}

// Column selection + alphabet lookup + validity in one pass: out[r][k] = lut[mat[r][cols[k]]] (lut: 256 entries, -1 =
// outside the alphabet), valid[r] = no -1 in the row.  Rows are written for every r (invalid ones included); the caller
// keeps the valid ones.
extern "C" int plm_encode_columns(const uint8_t *mat, int64_t n_rows, int64_t width, const int64_t *cols, int64_t n_cols,
                                  const int8_t *lut256, int8_t *out, uint8_t *valid) {
    if (!mat || !cols || !lut256 || !out || !valid) return plm_fail(PLM_EINVAL, "plm_encode_columns: NULL argument");
    for (int64_t k = 0; k < n_cols; k++)
        if (cols[k] < 0 || cols[k] >= width) return plm_fail(PLM_EINVAL, "plm_encode_columns: column outside the alignment");
    for (int64_t r = 0; r < n_rows; r++) {
        const uint8_t *row = mat + r * width;
        int8_t *o = out + r * n_cols;
        int bad = 0;
        for (int64_t k = 0; k < n_cols; k++) {
            const int8_t v = lut256[row[cols[k]]];
            o[k] = v;
            bad |= v < 0;
        }
        valid[r] = bad ? 0 : 1;
    }
    return PLM_OK;
}

// The raw EC file (couplings/pairs.py:55-58 reads it; plmc writes it): one line per site pair i < j, i ascending then j,
// "index_i A_i index_j A_j 0 cn" with cn printed "%.6f".  44 850 lines at L = 300 were 20-55 ms of Python string formatting
// in every run_plmc_hip call; here one buffer and one fwrite.  cn: dense row-major [L][L] doubles (the caller's float32
// scores widened exactly as Python's "%.6f" % numpy.float32 does).  model_io.write_raw_ec_file keeps its Python twin (fallback
// and test oracle: both reproduce the reference's two real plmc outputs byte for byte).
extern "C" int plm_write_raw_ec_file(const char *path, int32_t n_sites, const int32_t *index_list, const char *target_seq,
                                     const double *cn) {
    if (!path || n_sites < 0 || !index_list || !target_seq || !cn) return plm_fail(PLM_EINVAL, "plm_write_raw_ec_file: NULL argument");
    FILE *f = fopen(path, "wb");
    if (!f) return plm_fail(PLM_EINVAL, "cannot open %s for writing", path);
    std::vector<char> buf;
    buf.reserve((size_t)1 << 20);
    char line[128];
    bool ok = true;
    for (int32_t i = 0; i < n_sites && ok; i++) {
        for (int32_t j = i + 1; j < n_sites; j++) {
            const int n = snprintf(line, sizeof line, "%d %c %d %c 0 %.6f\n", index_list[i], target_seq[i], index_list[j],
                                   target_seq[j], cn[(size_t)i * n_sites + j]);
            if (n <= 0 || n >= (int)sizeof line) { ok = false; break; }
            buf.insert(buf.end(), line, line + n);
        }
        if (buf.size() >= ((size_t)1 << 20) - 4096 || i == n_sites - 1) {
            if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) ok = false;
            buf.clear();
        }
    }
    if (n_sites < 2 && ok) ok = fputc('\n', f) != EOF;      // (the Python twin ends an empty table with one newline)
    if (fclose(f) != 0) ok = false;
    return ok ? PLM_OK : plm_fail(PLM_EINVAL, "writing %s failed", path);
}
