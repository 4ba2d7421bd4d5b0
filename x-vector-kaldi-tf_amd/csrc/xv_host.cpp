// xv_host.cpp -- host-side (CPU) helpers of the extraction path: libxvector_host.so.
//
// xv_ark_scan_fm: one pass over a memory block of a Kaldi ark stream, returning the layout of every COMPLETE
// binary float-matrix record ("<key> \0BFM \4<int32 rows>\4<int32 cols><rows*cols float32>").  This replaces the
// per-byte / per-field Python parsing of the reference's reader (local/tf/kaldi_io.py:120-133, 395-437) for the
// record type the feature pipeline emits; anything else (DM, CM, text) stops the scan and is left to the generic
// Python reader.
#include <stddef.h>
#include <stdint.h>
#include <string.h>

extern "C" {

int xv_host_version(void) { return 2; }

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

// Copies the payloads of n scanned records (data_off[i], rows[i] x cols float32, all with the same column count) back to
// back into dst -- one GIL-free call instead of one NumPy copy per utterance.  Returns the number of rows copied.
int64_t xv_ark_gather_fm(const uint8_t *buf, const int64_t *data_off, const int32_t *rows, int cols, int n, float *dst)
{
    int64_t r = 0;
    for (int i = 0; i < n; ++i) {
        const size_t nbytes = (size_t)rows[i] * (size_t)cols * 4;
        memcpy(dst + (size_t)r * cols, buf + data_off[i], nbytes);
        r += rows[i];
    }
    return r;
}

}  // extern "C"
