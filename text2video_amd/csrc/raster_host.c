/* raster_host.c -- host-side inner loops of the pose-map rasteriser (text2video_amd/keypoints.py), plain C.
 *
 * The stamping rule of the reference's keypoint2img.py (drawEdge / setColor, :17-66; semantics in SURVEY App. D) is
 * sequential by construction: for every offset of the pen, "all pixels still black -> paint the colour, else average
 * with the colour" is decided for the WHOLE segment at once, and later offsets see what earlier ones wrote.  In numpy
 * that is ~40 fancy-indexing round trips per segment and ~100 segments per frame (20-45 ms per frame and core); here it
 * is the same arithmetic in two tight loops per offset.  Bit-identical to the numpy form: values are read for all points
 * first and written afterwards (duplicate pixels receive the same value), (cur + colour) / 2 is evaluated in double and
 * truncated, coordinates are clamped to the image.
 *
 * Built by csrc/Makefile into lib/libt2v_host.so (gcc; no HIP).  keypoints.py falls back to its numpy form if the
 * library is absent -- this is host glue, not the device path.
 */
#include <stdint.h>
#include <stdlib.h>

/* coordinates stay 64-bit until they are clamped (numpy clips int64 values): a far-off key point cannot overflow */
static inline long clampl(long v, long lo, long hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* one pen offset over n points: all-black test, then paint or average */
static void blend_points(uint8_t* img, int h, int w, const long* xs, const long* ys, int n, int ox, int oy,
                         const double rgb[3]) {
    int all_zero = 1;
    for (int i = 0; i < n && all_zero; ++i) {
        const uint8_t* p = img + (clampl(ys[i] + oy, 0, h - 1) * w + clampl(xs[i] + ox, 0, w - 1)) * 3;
        all_zero = (p[0] | p[1] | p[2]) == 0;
    }
    if (all_zero) {
        for (int i = 0; i < n; ++i) {
            uint8_t* p = img + (clampl(ys[i] + oy, 0, h - 1) * w + clampl(xs[i] + ox, 0, w - 1)) * 3;
            p[0] = (uint8_t)rgb[0]; p[1] = (uint8_t)rgb[1]; p[2] = (uint8_t)rgb[2];
        }
        return;
    }
    /* averages are formed from the values BEFORE this offset's writes (numpy reads the whole index vector first, then
     * assigns: points that map to one pixel all receive that pixel's OLD value averaged with the colour): two passes */
    enum { CH = 1024 };
    uint8_t stack_tmp[CH * 3];
    uint8_t* tmp = n <= CH ? stack_tmp : (uint8_t*)malloc((size_t)n * 3);
    if (!tmp) return;
    for (int i = 0; i < n; ++i) {
        const uint8_t* p = img + (clampl(ys[i] + oy, 0, h - 1) * w + clampl(xs[i] + ox, 0, w - 1)) * 3;
        for (int c = 0; c < 3; ++c) tmp[i * 3 + c] = (uint8_t)(((double)p[c] + rgb[c]) / 2.0);
    }
    for (int i = 0; i < n; ++i) {
        uint8_t* p = img + (clampl(ys[i] + oy, 0, h - 1) * w + clampl(xs[i] + ox, 0, w - 1)) * 3;
        p[0] = tmp[i * 3]; p[1] = tmp[i * 3 + 1]; p[2] = tmp[i * 3 + 2];
    }
    if (tmp != stack_tmp) free(tmp);
}

/* keypoints.stamp(img, xs, ys, bw, rgb, caps): pen offsets range(-bw, bw)^2 over the whole segment, then (caps) the
 * round end caps i*i + j*j < 4*bw*bw around the first and last point, both end points per offset at once */
void t2v_raster_stamp(uint8_t* img, int h, int w, const long* xs, const long* ys, int n, int bw, const double* rgb,
                      int caps) {
    if (n <= 0) return;
    for (int oy = -bw; oy < bw; ++oy)
        for (int ox = -bw; ox < bw; ++ox) blend_points(img, h, w, xs, ys, n, ox, oy, rgb);
    if (caps) {
        const long ex[2] = {xs[0], xs[n - 1]}, ey[2] = {ys[0], ys[n - 1]};
        for (int oy = -2 * bw; oy < 2 * bw; ++oy)
            for (int ox = -2 * bw; ox < 2 * bw; ++ox)
                if (oy * oy + ox * ox < 4 * bw * bw) blend_points(img, h, w, ex, ey, 2, ox, oy, rgb);
    }
}

int t2v_raster_abi(void) { return 1; }
