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
#include <cstring>

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
