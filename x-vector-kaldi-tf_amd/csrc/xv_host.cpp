// xv_host.cpp -- host-side (CPU) helpers of the extraction path: libxvector_host.so.
//
// xv_ark_scan_fm: one pass over a memory block of a Kaldi ark stream, returning the layout of every COMPLETE
// binary float-matrix record ("<key> \0BFM \4<int32 rows>\4<int32 cols><rows*cols float32>").  This replaces the
// per-byte / per-field Python parsing of the reference's reader (local/tf/kaldi_io.py:120-133, 395-437) for the
// record type the feature pipeline emits; anything else (DM, CM, text) stops the scan and is left to the generic
// Python reader.
//
// xv_pack_rows_f32: the batch packer of the extractor (xvector_amd/engine.py BatchLayout): chunk i = len[i] rows of
// feat_dim floats at address src[i] -> rows [dst_row[i], dst_row[i]+len[i]) of the (pinned) staging matrix, every other
// row zeroed, row_valid filled.  One GIL-free call on a few threads instead of one np.concatenate per batch.
//
// The other entry points, each described where it is defined: xv_ark_scan_fv (float-vector records: the VAD tables),
// xv_ark_index_fd (header-only index of an ark FILE: byte-range sharding across ranks), xv_ark_decode_cm (Kaldi's
// CompressedMatrix records decoded into arenas, Kaldi's float32 arithmetic), xv_ark_keys / xv_ark_gather_fm (keys and
// matrices of a scanned block gathered into one array each), xv_copy_bytes (a memcpy outside the interpreter lock),
// xv_raw_row_plan (destination row of every raw frame of a batch: the recipe's CMN + VAD mode), xv_scp_line_index (where the lines
// and keys of an scp table lie: a rank of a sharded job cuts its line range out of the text without splitting the other ranks'
// lines), xv_vec_records_write_fd (the x-vector table of a job, ark records + scp lines, written from the gathered block as it lies).
// An internal helper library of the Python host side (ctypes, local/tf/kaldi_io.py and xvector_amd/engine.py) -- the drop-in
// boundary of the compute path is libxvector_hip.so (include/xvector_hip.h).  Built for x86-64-v3 (AVX2).
#include <errno.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <immintrin.h>

#include <thread>
#include <vector>

extern "C" {

int xv_host_version(void) { return 11; }

// Index pass over an ark FILE of binary float-matrix records (plain "FM " or compressed "CM ") without reading the matrices: per
// record one pread of the header ("<key> \0BFM \4<rows>\4<cols>" / "<key> \0BCM <min><range><rows><cols>"), then a hop over the payload.  This is what lets the ranks of a job split a
// seekable ark by BYTE RANGES (local/tf/models.py make_embedding): the format has no sync marks, so record boundaries can only
// be found from the front -- but finding them costs one small read per record (~1 us), not a parse of the 27 KB behind it.
// rec_off[i] = offset of record i (its key), rows[i] / cols[i] its shape; keys (optional) receives the keys back to back with
// '\n' separators, *keys_used the bytes written.  Returns the number of records; *next = offset behind the last one taken;
// *stop = 0 end of range reached, 1 a record that is not a complete binary FM record (caller falls back), 2 max_records or the
// key buffer is full (call again from *next).
int64_t xv_ark_index_fd(int fd, int64_t pos, int64_t end, int64_t max_records, int64_t *rec_off, int32_t *rows, int32_t *cols,
                        uint8_t *keys, int64_t keys_cap, int64_t *keys_used, int64_t *next, int *stop)
{
    int64_t n = 0, kw = 0;
    *stop = 0;
    uint8_t hdr[4096];
    while (pos < end) {
        if (n == max_records) { *stop = 2; break; }
        size_t want = 320;
        ssize_t got = pread(fd, hdr, want, (off_t)pos);
        if (got <= 0) { *stop = 1; break; }
        const uint8_t *sp = (const uint8_t *)memchr(hdr, ' ', (size_t)got);
        if (!sp && got == (ssize_t)want) {                         // a very long key: look again with the large buffer
            got = pread(fd, hdr, sizeof(hdr), (off_t)pos);
            sp = got > 0 ? (const uint8_t *)memchr(hdr, ' ', (size_t)got) : nullptr;
        }
        if (!sp) { *stop = 1; break; }
        const size_t klen = (size_t)(sp - hdr), h = klen + 1;
        if (h + 15 > (size_t)got) {
            if (pos + (int64_t)h + 15 > end) { *stop = 1; break; } // truncated header
            got = pread(fd, hdr, sizeof(hdr), (off_t)pos);
            if (got < (ssize_t)(h + 15)) { *stop = 1; break; }
        }
        int32_t r, c;
        int64_t after;
        if (hdr[h] == 0 && hdr[h + 1] == 'B' && hdr[h + 2] == 'C' && hdr[h + 3] == 'M' &&
            (hdr[h + 4] == ' ' || ((hdr[h + 4] == '2' || hdr[h + 4] == '3') && hdr[h + 5] == ' '))) {
            // compressed matrix: {float min, float range, int32 rows, int32 cols}, then "CM ": cols x 8 header bytes + rows x cols
            // bytes; "CM2" / "CM3": rows x cols uint16 / uint8
            const int kind = hdr[h + 4] == ' ' ? 1 : hdr[h + 4] - '0';
            const size_t tag = kind == 1 ? 5 : 6;
            if (h + tag + 16 > (size_t)got) {
                got = pread(fd, hdr, sizeof(hdr), (off_t)pos);
                if (got < (ssize_t)(h + tag + 16)) { *stop = 1; break; }
            }
            memcpy(&r, hdr + h + tag + 8, 4);
            memcpy(&c, hdr + h + tag + 12, 4);
            if (r < 0 || c < 0) { *stop = 1; break; }
            // (payload checked against the range BEFORE it is added: r x c near 2^31 x 2^31 times the element size leaves int64 and
            // "after" would move backwards)
            const uint64_t room = (uint64_t)(end - pos), esz = kind == 2 ? 2 : 1;
            // (only "CM " carries the 8-byte per-column headers; CM2 / CM3 -- what Kaldi writes for matrices of < 8 rows -- are r c
            // elements and nothing else: a short CM2 record at the end of the range is legitimate)
            if ((kind == 1 && (uint64_t)c * 8 > room) || (c != 0 && (uint64_t)r > (room / esz) / (uint64_t)c)) { *stop = 1; break; }
            after = pos + (int64_t)h + (int64_t)tag + 16 +
                    (kind == 1 ? (int64_t)c * 8 + (int64_t)r * (int64_t)c : (int64_t)r * (int64_t)c * (int64_t)esz);
        } else {
            if (hdr[h] != 0 || hdr[h + 1] != 'B' || hdr[h + 2] != 'F' || hdr[h + 3] != 'M' || hdr[h + 4] != ' ' || hdr[h + 5] != 4 ||
                hdr[h + 10] != 4) { *stop = 1; break; }
            memcpy(&r, hdr + h + 6, 4);
            memcpy(&c, hdr + h + 11, 4);
            if (r < 0 || c < 0) { *stop = 1; break; }
            if (c != 0 && (uint64_t)r > ((uint64_t)(end - pos) / 4) / (uint64_t)c) { *stop = 1; break; }
            after = pos + (int64_t)h + 15 + (int64_t)r * (int64_t)c * 4;
        }
        if (after > end || after <= pos) { *stop = 1; break; }     // truncated payload
        if (keys) {
            if (memchr(hdr, '\n', klen)) { *stop = 1; break; }     // keys come back newline-separated: such a key cannot
            if (kw + (int64_t)klen + 1 > keys_cap) { *stop = 2; break; }
            memcpy(keys + kw, hdr, klen);
            kw += (int64_t)klen;
            keys[kw++] = '\n';
        }
        rec_off[n] = pos;
        rows[n] = r;
        cols[n] = c;
        ++n;
        pos = after;
    }
    if (keys_used) *keys_used = kw;
    *next = pos;
    return n;
}

// memcpy callable through ctypes, i.e. WITHOUT the interpreter lock: the in-place reader fills its arenas from an in-memory
// stream (io.BytesIO) with it -- BytesIO.readinto copies under the lock, and 64 MB at a time stalls every other thread of the
// extraction pipeline (model load, planning, packing) for milliseconds.
void xv_copy_bytes(void *dst, const void *src, size_t n) { memcpy(dst, src, n); }

// Scans buf[pos, len).  Fills up to max_records entries; returns the number of records found.
// *next = offset of the first byte not consumed; *stop = 0 buffer exhausted / record incomplete (need more data),
// 1 next record is not a binary FM record (caller handles one record generically), 2 max_records reached.
int xv_ark_scan_fm(const uint8_t *buf, size_t pos, size_t len, int max_records, int64_t *key_off, int32_t *key_len,
                   int64_t *data_off, int32_t *rows, int32_t *cols, size_t *next, int *stop)
{
    int n = 0;
    *stop = 0;
    while (n < max_records) {
        const uint8_t *sp = (const uint8_t *)memchr(buf + pos, ' ', len - pos);
        if (!sp) break;                                            // key not complete yet
        const size_t kend = (size_t)(sp - buf);
        const size_t h = kend + 1;                                 // "\0B" "FM " \4 rows \4 cols  = 2 + 3 + 10 bytes
        if (h + 15 > len) break;
        if (buf[h] != 0 || buf[h + 1] != 'B' || buf[h + 2] != 'F' || buf[h + 3] != 'M' || buf[h + 4] != ' ' ||
            buf[h + 5] != 4 || buf[h + 10] != 4) {
            *stop = 1;
            break;
        }
        int32_t r, c;
        memcpy(&r, buf + h + 6, 4);
        memcpy(&c, buf + h + 11, 4);
        if (r < 0 || c < 0) { *stop = 1; break; }
        const size_t d = h + 15;
        const size_t nbytes = (size_t)r * (size_t)c * 4;
        if (d + nbytes > len) break;                               // payload incomplete
        key_off[n] = (int64_t)pos;
        key_len[n] = (int32_t)(kend - pos);
        data_off[n] = (int64_t)d;
        rows[n] = r;
        cols[n] = c;
        ++n;
        pos = d + nbytes;
    }
    if (n == max_records) *stop = 2;
    *next = pos;
    return n;
}

// Kaldi CompressedMatrix records of the speech-feature kind ("<key> \0BCM " + {float min, float range, int32 rows, int32 cols} +
// cols x {uint16 percentile 0, 25, 75, 100} + cols x rows bytes, COLUMN-major): what steps/make_mfcc.sh writes by default, i.e. what a
// feats.scp points at when the extractor reads the features itself (its own CMN / VAD front-end) instead of being fed by Kaldi's
// pipes.  Scans buf[pos, len) for consecutive complete CM records with the column count of the first one, decodes them -- the
// float32 arithmetic of local/tf/kaldi_io.py:455-502, which is Kaldi's, operation for operation (this file is built with
// -ffp-contract=off) -- into `out` as row-major float32 matrices, on `nthreads` threads.  out_off[i] = byte offset of matrix i
// in out.  *stop: 0 buffer exhausted / record incomplete, 1 the next record is not such a record, 2 max_records reached, 3 out is
// full, 4 the column count changed.
static inline float cm_u16(float gmin, float grange, uint16_t v) { return gmin + grange * 1.52590218966964e-05f * (float)v; }

static void cm_decode_one(const uint8_t *rec, int rows, int cols, float *out, std::vector<float> &tmp, int kind)
{
    float gmin, grange;
    memcpy(&gmin, rec, 4);
    memcpy(&grange, rec + 4, 4);
    if (kind != 1) {
        // "CM2" (row-major uint16) / "CM3" (row-major uint8): what Kaldi's automatic method writes for matrices of <= 8 rows
        const size_t n = (size_t)rows * cols;
        if (kind == 2) {
            const float step2 = grange * 1.52590218966964e-05f;
            const uint8_t *d = rec + 16;
            for (size_t i = 0; i < n; ++i) {
                uint16_t v;
                memcpy(&v, d + 2 * i, 2);
                out[i] = gmin + step2 * (float)v;
            }
        } else {
            const float step3 = grange * (float)(1.0 / 255.0);
            const uint8_t *d = rec + 16;
            for (size_t i = 0; i < n; ++i) out[i] = gmin + step3 * (float)d[i];
        }
        return;
    }
    const uint8_t *hdr = rec + 16;
    const uint8_t *data = hdr + (size_t)cols * 8;
    const float step = grange * 1.52590218966964e-05f;            // (evaluated left to right, as NumPy does: (range * step) * pct)
    if (rows >= 8 && cols >= 8) {
        // 8 rows x 8 columns at a time, in registers (AVX2; this file is built for x86-64-v3): eight bytes of each of eight columns
        // -> float, the column's three line segments (separate multiply and add: no contraction, the same float32 operations as the
        // scalar form below), an 8 x 8 transpose, eight 32-byte row stores.  The last block of rows / columns overlaps its
        // neighbour instead of falling back to scalar code (it writes the same values again).
        float cst[6][256];
        std::vector<float> big;
        float *P0 = cst[0], *A = cst[1], *P25 = cst[2], *B = cst[3], *P75 = cst[4], *D = cst[5];
        if (cols > 256) {
            big.resize((size_t)6 * cols);
            P0 = big.data(); A = P0 + cols; P25 = A + cols; B = P25 + cols; P75 = B + cols; D = P75 + cols;
        }
        for (int c = 0; c < cols; ++c) {
            uint16_t q[4];
            memcpy(q, hdr + (size_t)c * 8, 8);
            const float p0 = gmin + step * (float)q[0], p25 = gmin + step * (float)q[1], p75 = gmin + step * (float)q[2],
                        p100 = gmin + step * (float)q[3];
            P0[c] = p0; P25[c] = p25; P75[c] = p75;
            A[c] = (p25 - p0) / 64.f; B[c] = (p75 - p25) / 128.f; D[c] = (p100 - p75) / 63.f;
        }
        const __m256 f64 = _mm256_set1_ps(64.f), f192 = _mm256_set1_ps(192.f);
        const __m256i i64 = _mm256_set1_epi32(64), i192 = _mm256_set1_epi32(192);
        for (int rb = 0; rb < rows; rb += 8) {
            const int r0 = rb + 8 <= rows ? rb : rows - 8;
            for (int cb = 0; cb < cols; cb += 8) {
                const int c0 = cb + 8 <= cols ? cb : cols - 8;
                __m256 v[8];
                for (int j = 0; j < 8; ++j) {
                    const int c = c0 + j;
                    const __m256i ui = _mm256_cvtepu8_epi32(_mm_loadl_epi64((const __m128i *)(data + (size_t)c * rows + r0)));
                    const __m256 uf = _mm256_cvtepi32_ps(ui);
                    const __m256 lo = _mm256_add_ps(_mm256_broadcast_ss(P0 + c), _mm256_mul_ps(_mm256_broadcast_ss(A + c), uf));
                    const __m256 mid = _mm256_add_ps(_mm256_broadcast_ss(P25 + c), _mm256_mul_ps(_mm256_broadcast_ss(B + c), _mm256_sub_ps(uf, f64)));
                    const __m256 hi = _mm256_add_ps(_mm256_broadcast_ss(P75 + c), _mm256_mul_ps(_mm256_broadcast_ss(D + c), _mm256_sub_ps(uf, f192)));
                    const __m256 x = _mm256_blendv_ps(lo, mid, _mm256_castsi256_ps(_mm256_cmpgt_epi32(ui, i64)));
                    v[j] = _mm256_blendv_ps(x, hi, _mm256_castsi256_ps(_mm256_cmpgt_epi32(ui, i192)));
                }
                const __m256 t0 = _mm256_unpacklo_ps(v[0], v[1]), t1 = _mm256_unpackhi_ps(v[0], v[1]), t2 = _mm256_unpacklo_ps(v[2], v[3]),
                             t3 = _mm256_unpackhi_ps(v[2], v[3]), t4 = _mm256_unpacklo_ps(v[4], v[5]), t5 = _mm256_unpackhi_ps(v[4], v[5]),
                             t6 = _mm256_unpacklo_ps(v[6], v[7]), t7 = _mm256_unpackhi_ps(v[6], v[7]);
                const __m256 s0 = _mm256_shuffle_ps(t0, t2, 0x44), s1 = _mm256_shuffle_ps(t0, t2, 0xEE), s2 = _mm256_shuffle_ps(t1, t3, 0x44),
                             s3 = _mm256_shuffle_ps(t1, t3, 0xEE), s4 = _mm256_shuffle_ps(t4, t6, 0x44), s5 = _mm256_shuffle_ps(t4, t6, 0xEE),
                             s6 = _mm256_shuffle_ps(t5, t7, 0x44), s7 = _mm256_shuffle_ps(t5, t7, 0xEE);
                float *o = out + (size_t)r0 * cols + c0;
                _mm256_storeu_ps(o, _mm256_permute2f128_ps(s0, s4, 0x20));
                _mm256_storeu_ps(o + (size_t)cols, _mm256_permute2f128_ps(s1, s5, 0x20));
                _mm256_storeu_ps(o + (size_t)2 * cols, _mm256_permute2f128_ps(s2, s6, 0x20));
                _mm256_storeu_ps(o + (size_t)3 * cols, _mm256_permute2f128_ps(s3, s7, 0x20));
                _mm256_storeu_ps(o + (size_t)4 * cols, _mm256_permute2f128_ps(s0, s4, 0x31));
                _mm256_storeu_ps(o + (size_t)5 * cols, _mm256_permute2f128_ps(s1, s5, 0x31));
                _mm256_storeu_ps(o + (size_t)6 * cols, _mm256_permute2f128_ps(s2, s6, 0x31));
                _mm256_storeu_ps(o + (size_t)7 * cols, _mm256_permute2f128_ps(s3, s7, 0x31));
            }
        }
        return;
    }
    // small matrices -- pass 1: column by column as the bytes lie (contiguous loads and stores: the compiler vectorises it); pass 2: transpose
    if (tmp.size() < (size_t)rows * cols) tmp.resize((size_t)rows * cols);
    float *t = tmp.data();
    for (int c = 0; c < cols; ++c) {
        uint16_t q[4];
        memcpy(q, hdr + (size_t)c * 8, 8);
        const float p0 = gmin + step * (float)q[0], p25 = gmin + step * (float)q[1], p75 = gmin + step * (float)q[2],
                    p100 = gmin + step * (float)q[3];
        const float a = (p25 - p0) / 64.f, b = (p75 - p25) / 128.f, d = (p100 - p75) / 63.f;
        const uint8_t *u = data + (size_t)c * rows;
        float *o = t + (size_t)c * rows;
        for (int r = 0; r < rows; ++r) {
            const float uf = (float)u[r];
            const float lo = p0 + a * uf, mid = p25 + b * (uf - 64.f), hi = p75 + d * (uf - 192.f);
            o[r] = u[r] <= 64 ? lo : (u[r] <= 192 ? mid : hi);
        }
    }
    for (int r = 0; r < rows; ++r) {
        float *o = out + (size_t)r * cols;
        const float *ti = t + r;
        for (int c = 0; c < cols; ++c) o[c] = ti[(size_t)c * rows];
    }
}

int xv_ark_decode_cm(const uint8_t *buf, size_t pos, size_t len, int max_records, uint8_t *out, size_t out_cap, int64_t *key_off,
                     int32_t *key_len, int64_t *out_off, int32_t *rows, int32_t *cols, size_t *next, int *stop, int nthreads)
{
    int n = 0;
    size_t used = 0;
    int c0 = -1;
    *stop = 0;
    std::vector<const uint8_t *> recs;
    std::vector<int> kinds;
    while (true) {
        if (n == max_records) { *stop = 2; break; }
        const uint8_t *sp = (const uint8_t *)memchr(buf + pos, ' ', len - pos);
        if (!sp) break;
        const size_t kend = (size_t)(sp - buf);
        const size_t h = kend + 1;                                 // "\0B" "CM " + 16-byte global header
        if (h + 21 > len) break;
        if (buf[h] != 0 || buf[h + 1] != 'B' || buf[h + 2] != 'C' || buf[h + 3] != 'M' ||
            (buf[h + 4] != ' ' && buf[h + 4] != '2' && buf[h + 4] != '3') || (buf[h + 4] != ' ' && buf[h + 5] != ' ')) { *stop = 1; break; }
        const int kind = buf[h + 4] == ' ' ? 1 : buf[h + 4] - '0';                // "CM " | "CM2 " | "CM3 "
        const size_t tag = kind == 1 ? 5 : 6;
        if (h + tag + 16 > len) break;
        int32_t r, c;
        memcpy(&r, buf + h + tag + 8, 4);
        memcpy(&c, buf + h + tag + 12, 4);
        if (r < 0 || c < 0) { *stop = 1; break; }
        if (c0 >= 0 && c != c0) { *stop = 4; break; }
        const size_t d = h + tag;
        const size_t nbytes = 16 + (kind == 1 ? (size_t)c * 8 + (size_t)r * (size_t)c : (size_t)r * (size_t)c * (kind == 2 ? 2 : 1));
        if (d + nbytes > len) break;                               // payload incomplete
        const size_t need = ((size_t)r * (size_t)c * 4 + 63) & ~(size_t)63;
        if (used + need > out_cap) { *stop = 3; break; }
        c0 = c;
        key_off[n] = (int64_t)pos;
        key_len[n] = (int32_t)(kend - pos);
        out_off[n] = (int64_t)used;
        rows[n] = r;
        cols[n] = c;
        recs.push_back(buf + d);
        kinds.push_back(kind);
        used += need;
        ++n;
        pos = d + nbytes;
    }
    *next = pos;
    if (n > 0) {
        const int nt = nthreads < 1 ? 1 : (nthreads > n ? n : nthreads);
        auto work = [&](int t) {
            std::vector<float> tmp;
            for (int i = t; i < n; i += nt) cm_decode_one(recs[i], rows[i], cols[i], reinterpret_cast<float *>(out + out_off[i]), tmp, kinds[i]);
        };
        if (nt == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
        }
    }
    return n;
}

// The same scan for binary float VECTOR records ("<key> \0BFV \4<int32 dim><dim float32>", e.g. a vad.ark): reported as
// dim x 1 matrices (rows = dim, cols = 1), so that xv_ark_gather_fm gathers them too.
int xv_ark_scan_fv(const uint8_t *buf, size_t pos, size_t len, int max_records, int64_t *key_off, int32_t *key_len,
                   int64_t *data_off, int32_t *rows, int32_t *cols, size_t *next, int *stop)
{
    int n = 0;
    *stop = 0;
    while (n < max_records) {
        const uint8_t *sp = (const uint8_t *)memchr(buf + pos, ' ', len - pos);
        if (!sp) break;
        const size_t kend = (size_t)(sp - buf);
        const size_t h = kend + 1;                                 // "\0B" "FV " \4 dim = 2 + 3 + 5 bytes
        if (h + 10 > len) break;
        if (buf[h] != 0 || buf[h + 1] != 'B' || buf[h + 2] != 'F' || buf[h + 3] != 'V' || buf[h + 4] != ' ' || buf[h + 5] != 4) {
            *stop = 1;
            break;
        }
        int32_t dim;
        memcpy(&dim, buf + h + 6, 4);
        if (dim < 0) { *stop = 1; break; }
        const size_t d = h + 10;
        const size_t nbytes = (size_t)dim * 4;
        if (d + nbytes > len) break;
        key_off[n] = (int64_t)pos;
        key_len[n] = (int32_t)(kend - pos);
        data_off[n] = (int64_t)d;
        rows[n] = dim;
        cols[n] = 1;
        ++n;
        pos = d + nbytes;
    }
    if (n == max_records) *stop = 2;
    *next = pos;
    return n;
}

// The keys of n scanned records, whitespace-stripped, validated against Kaldi's key alphabet [./a-zA-Z0-9_-]+ and written
// back to back with '\n' separators (no trailing separator): ONE decode + split in Python instead of a slice, a decode, a
// strip and a regular-expression match per utterance (that loop held the interpreter lock for milliseconds per window).
// Returns the number of bytes written, or -(i+1) when key i is empty / malformed / does not fit (the caller then takes the
// per-key path, which reports it).
int64_t xv_ark_keys(const uint8_t *buf, const int64_t *key_off, const int32_t *key_len, int n, uint8_t *out, int64_t cap)
{
    int64_t w = 0;
    for (int i = 0; i < n; ++i) {
        const uint8_t *k = buf + key_off[i];
        int32_t len = key_len[i];
        while (len > 0 && (*k == ' ' || *k == '\n' || *k == '\r' || *k == '\t')) { ++k; --len; }
        while (len > 0 && (k[len - 1] == ' ' || k[len - 1] == '\n' || k[len - 1] == '\r' || k[len - 1] == '\t')) --len;
        if (len <= 0 || w + len + 1 > cap) return -(int64_t)(i + 1);
        for (int32_t j = 0; j < len; ++j) {
            const uint8_t c = k[j];
            const bool ok = (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == '.' || c == '/' ||
                            c == '_' || c == '-';
            if (!ok) return -(int64_t)(i + 1);
            out[w + j] = c;
        }
        w += len;
        if (i + 1 < n) out[w++] = '\n';
    }
    return w;
}

// Copies the payloads of n scanned records (data_off[i], rows[i] x cols float32, all with the same column count) back to
// back into dst -- one GIL-free call instead of one NumPy copy per utterance.  Returns the number of rows copied.
int64_t xv_ark_gather_fm(const uint8_t *buf, const int64_t *data_off, const int32_t *rows, int cols, int n, float *dst)
{
    std::vector<int64_t> first(n + 1, 0);                          // destination row of every record
    for (int i = 0; i < n; ++i) first[i + 1] = first[i] + rows[i];
    auto work = [&](int i0, int i1) {
        for (int i = i0; i < i1; ++i)
            memcpy(dst + (size_t)first[i] * cols, buf + data_off[i], (size_t)rows[i] * (size_t)cols * 4);
    };
    const size_t total = (size_t)first[n] * (size_t)cols * 4;
    const int nt = total >= ((size_t)8 << 20) && n >= 64 ? 4 : 1;   // a scanner pass of 8192 utterances is ~230 MB
    if (nt == 1) {
        work(0, n);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t) pool.emplace_back(work, (int)((int64_t)n * t / nt), (int)((int64_t)n * (t + 1) / nt));
        for (auto &th : pool) th.join();
    }
    return first[n];
}

// xv_raw_row_plan: where every RAW frame of a batch's utterances goes (Extractor.submit_raw of xvector_amd/engine.py: sliding CMN +
// select-voiced-frames on the device, extract_xvectors.sh:68).  Utterance i brings T[i] frames; its voiced flags are voiced + vstart[i]
// (one byte per frame, non-zero = voiced; voiced == NULL: every frame).  Its voiced frames, counted from 0, fall into chunks of
// size[i] frames; the first kept[i] of them exist, and chunk k of the utterance is chunk seg[i] + k of the window -- of which this
// batch holds chunks [b0, b1), chunk c at row row_start[c - b0].  dst[frame] = row of the packed batch, or -1 (unvoiced, a dropped
// tail, a chunk of another batch).  One pass over the frames instead of a dozen NumPy passes over 8-byte temporaries.
// Returns 0, or -1 when the frames do not fit dst_cap.
int xv_raw_row_plan(int n_utt, const int64_t *T, const uint8_t *voiced, const int64_t *vstart, const int64_t *size, const int64_t *kept,
                    const int64_t *seg, int64_t b0, int64_t b1, const int32_t *row_start, int32_t *dst, int64_t dst_cap)
{
    int64_t o = 0;
    for (int i = 0; i < n_utt; ++i) {
        const int64_t Ti = T[i];
        if (Ti < 0 || o + Ti > dst_cap) return -1;
        const uint8_t *fl = voiced ? voiced + vstart[i] : nullptr;
        const int64_t sz = size[i], kp = sz > 0 ? kept[i] : 0, sg = seg[i];
        int32_t *d = dst + o;
        int64_t k = 0, r = 0;                               // chunk of the utterance, frame inside it
        for (int64_t t = 0; t < Ti; ++t) {
            if (fl && !fl[t]) { d[t] = -1; continue; }
            const int64_t cid = sg + k;
            d[t] = (k < kp && cid >= b0 && cid < b1) ? row_start[cid - b0] + (int32_t)r : -1;
            if (++r == sz) { r = 0; ++k; }
        }
        o += Ti;
    }
    return 0;
}

// Chunks must be given in ascending, non-overlapping dst_row order.  Columns [feat_dim, dst_ld) of dst are not touched (the
// caller keeps them zero).  Returns 0, or -1 on inconsistent arguments (nothing is written then).
int xv_pack_rows_f32(const uint64_t *src, const int32_t *len, const int32_t *dst_row, int n, int feat_dim, float *dst,
                     int64_t dst_ld, int64_t dst_rows, uint8_t *row_valid, int n_threads)
{
    if (n < 0 || feat_dim <= 0 || dst_ld < feat_dim || dst_rows < 0 || !dst) return -1;
    int64_t prev = 0;
    for (int i = 0; i < n; ++i) {
        if (len[i] < 0 || dst_row[i] < prev) return -1;
        prev = (int64_t)dst_row[i] + len[i];
    }
    if (prev > dst_rows) return -1;
    const size_t row_bytes = (size_t)feat_dim * 4;
    auto zero_rows = [&](int64_t a, int64_t b) {            // rows [a, b): feature columns and validity flags
        if (row_valid && b > a) memset(row_valid + a, 0, (size_t)(b - a));
        if (dst_ld == feat_dim) {
            if (b > a) memset(dst + a * dst_ld, 0, (size_t)(b - a) * row_bytes);
        } else {
            for (int64_t r = a; r < b; ++r) memset(dst + r * dst_ld, 0, row_bytes);
        }
    };
    auto work = [&](int i0, int i1, bool last) {            // chunks [i0, i1) and the gap in front of each of them
        for (int i = i0; i < i1; ++i) {
            const int64_t start = dst_row[i];
            zero_rows(i == 0 ? 0 : (int64_t)dst_row[i - 1] + len[i - 1], start);
            const float *s = reinterpret_cast<const float *>(src[i]);
            if (dst_ld == feat_dim) {
                memcpy(dst + start * dst_ld, s, (size_t)len[i] * row_bytes);
            } else {
                float *d = dst + start * dst_ld;
                for (int32_t r = 0; r < len[i]; ++r) memcpy(d + (size_t)r * dst_ld, s + (size_t)r * feat_dim, row_bytes);
            }
            if (row_valid) memset(row_valid + start, 1, (size_t)len[i]);
        }
        if (last) zero_rows(n == 0 ? 0 : (int64_t)dst_row[n - 1] + len[n - 1], dst_rows);
    };
    int nt = n_threads < 1 ? 1 : (n_threads > 16 ? 16 : n_threads);
    if (n < 4 * nt) nt = 1;
    if (nt == 1) {
        work(0, n, true);
        return 0;
    }
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; ++t) {
        const int i0 = (int)((int64_t)n * t / nt), i1 = (int)((int64_t)n * (t + 1) / nt);
        pool.emplace_back(work, i0, i1, t == nt - 1);
    }
    for (auto &th : pool) th.join();
    return 0;
}

// Where the non-blank lines of an scp table ("<key> <rxfilename>\n", extract_xvectors.sh:63-65 splits such a file per job) lie in
// its text: start[i] = offset of line i's first byte, end[i] = offset behind its last byte (the '\n' excluded), key_len[i] = bytes up to
// the first blank of the line (leading blanks belong to no key: such a line is reported as unusual).  One memchr-speed pass instead
// of decode + splitlines + split of every line on every rank of a job (1 M lines: 0.5 s of Python per rank).  Returns the number of
// lines (writes at most cap of them: call with cap = 0 to count), or -2 when the text holds a byte Python's str.splitlines would
// ALSO break lines at or that is not ASCII (\r, \v, \f, \x1c-\x1e, >= 0x80): the caller then takes the Python path, whose
// splitting rule is the documented one.
int64_t xv_scp_line_index(const uint8_t *buf, int64_t len, int64_t *start, int64_t *end, int32_t *key_len, int64_t cap)
{
    int64_t n = 0, pos = 0;
    while (pos < len) {
        const uint8_t *nl = static_cast<const uint8_t *>(memchr(buf + pos, '\n', (size_t)(len - pos)));
        const int64_t stop = nl ? (int64_t)(nl - buf) : len;
        bool blank = true;
        int32_t klen = -1;
        for (int64_t i = pos; i < stop; ++i) {
            const uint8_t c = buf[i];
            if (c >= 0x80 || c == '\r' || c == 0x0b || c == 0x0c || (c >= 0x1c && c <= 0x1e)) return -2;
            const bool ws = c == ' ' || c == '\t';
            if (!ws) blank = false;
            else if (klen < 0 && !blank) klen = (int32_t)(i - pos);
        }
        if (!blank) {
            if (buf[pos] == ' ' || buf[pos] == '\t') return -2;   // a line that starts with a blank: left to the Python path
            if (n < cap) {
                start[n] = pos;
                end[n] = stop;
                key_len[n] = klen < 0 ? (int32_t)(stop - pos) : klen;
            }
            ++n;
        }
        pos = stop + 1;
    }
    return n;
}

namespace {

bool write_all(int fd, const uint8_t *p, size_t n)
{
    while (n) {
        const ssize_t w = write(fd, p, n);
        if (w < 0) {
            if (errno == EINTR) continue;
            return false;
        }
        p += w;
        n -= (size_t)w;
    }
    return true;
}

inline size_t put_u64(uint8_t *dst, uint64_t v)
{
    uint8_t tmp[20];
    size_t k = 0;
    do { tmp[k++] = (uint8_t)('0' + v % 10); v /= 10; } while (v);
    for (size_t i = 0; i < k; ++i) dst[i] = tmp[k - 1 - i];
    return k;
}

}  // namespace

// The x-vector table of a job written from the block it was gathered into: for every row i of n with emitted[i] != 0 (all rows when
// emitted is NULL) the ark record "<key> " + "\0B" + "FV " + "\4" + uint32 dim + dim little-endian float32 -- the bytes of
// write_vec_flt (local/tf/kaldi_io.py:339-372 of the reference) -- goes to ark_fd and the line "<key> <ark_name>:<offset of the
// \0>\n" (what copy-vector ark,scp: prints, extract_xvectors.sh:74-88) to scp_fd (skipped when scp_fd < 0).  Key i = key_len[i]
// bytes at keys + key_off[i] (the scp text of the job's input, or keys joined in Python); vector i = dim floats at
// vecs + i * row_stride (the gathered [emitted? | x-vector] rows are read in place: row_stride = dim + 1, vecs = block + 1).
// pos0 = the ark's length so far.  Records are assembled in 4 MB pieces, one write(2) each: no second and third copy of a 2 GB
// table (NumPy row assembly + tobytes), no per-record Python.  Returns the number of records written and *pos_out = the ark's new
// length; -errno when a write fails.
int64_t xv_vec_records_write_fd(int ark_fd, int scp_fd, const uint8_t *keys, const int64_t *key_off, const int32_t *key_len, int64_t n,
                                const float *vecs, int dim, int64_t row_stride, const uint8_t *emitted, const char *ark_name,
                                int64_t pos0, int64_t *pos_out)
{
    const size_t cap = 4u << 20;
    const size_t name_len = strlen(ark_name);
    const size_t body = 10 + 4 * (size_t)dim;                       // "\0BFV " + "\4" + dim + payload
    std::vector<uint8_t> ark(cap + body + 4096), scp(scp_fd >= 0 ? cap / 4 + name_len + 4096 : 0);
    size_t aw = 0, sw = 0;
    int64_t pos = pos0, written = 0;
    const uint32_t d32 = (uint32_t)dim;
    for (int64_t i = 0; i < n; ++i) {
        if (emitted && !emitted[i]) continue;
        const size_t kl = (size_t)key_len[i];
        if (aw + kl + 1 + body > cap) {
            if (!write_all(ark_fd, ark.data(), aw)) return -(int64_t)errno;
            aw = 0;
            if (kl + 1 + body > ark.size()) ark.resize(kl + 1 + body);
        }
        const uint8_t *k = keys + key_off[i];
        uint8_t *a = ark.data() + aw;
        if (kl) {                                                    // (write_vec_flt: an empty key writes no key and no blank)
            memcpy(a, k, kl);
            a[kl] = ' ';
            a += kl + 1;
            pos += (int64_t)kl + 1;
        }
        if (scp_fd >= 0) {
            if (sw + kl + name_len + 24 > scp.size() - 64) {
                if (!write_all(scp_fd, scp.data(), sw)) return -(int64_t)errno;
                sw = 0;
                if (kl + name_len + 128 > scp.size()) scp.resize(kl + name_len + 4096);
            }
            uint8_t *t = scp.data() + sw;
            memcpy(t, k, kl); t += kl;
            *t++ = ' ';
            memcpy(t, ark_name, name_len); t += name_len;
            *t++ = ':';
            t += put_u64(t, (uint64_t)pos);
            *t++ = '\n';
            sw = (size_t)(t - scp.data());
        }
        a[0] = 0; a[1] = 'B'; a[2] = 'F'; a[3] = 'V'; a[4] = ' '; a[5] = 4;
        memcpy(a + 6, &d32, 4);
        memcpy(a + 10, vecs + i * row_stride, 4 * (size_t)dim);
        aw = (size_t)(a + body - ark.data());
        pos += (int64_t)body;
        ++written;
    }
    if (aw && !write_all(ark_fd, ark.data(), aw)) return -(int64_t)errno;
    if (sw && !write_all(scp_fd, scp.data(), sw)) return -(int64_t)errno;
    if (pos_out) *pos_out = pos;
    return written;
}

}  // extern "C"
