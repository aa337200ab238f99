// Host-side geometry of the Rectified-SpaAttn token order: generalized Hilbert ("Gilbert") curve over a
// t x h x w latent cuboid and the 26-neighbourhood relation between 128-token blocks along that curve.
// One-off start-up work (the reference spends 3 s + 3.5 s in per-point Python recursion at 32x45x80,
// utils/jenga_gilbert.py:84-288, :458-504, :613-693); here the curve is ENUMERATED once in curve order
// (the generator form of the same published recursion, J. Cerveny's gilbert3d), ~1 ms.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "rsa.h"

namespace {

struct V3 { int x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator-(V3 a) { return {-a.x, -a.y, -a.z}; }
inline int sgn(int v) { return v < 0 ? -1 : (v > 0 ? 1 : 0); }
inline int fdiv2(int v) { return v >= 0 ? v / 2 : -((-v + 1) / 2); }  // Python's floor division by 2
inline V3 half(V3 a) { return {fdiv2(a.x), fdiv2(a.y), fdiv2(a.z)}; }
inline V3 unit(V3 a) { return {sgn(a.x), sgn(a.y), sgn(a.z)}; }
inline int len(V3 a) { return abs(a.x + a.y + a.z); }

struct Emitter {
    int32_t* order;  // curve index -> linear index
    long n = 0;
    int h, w;        // linear = z*h*w + y*w + x
    void put(V3 p) { order[n++] = (int32_t)((long)p.z * h * w + (long)p.y * w + p.x); }
};

// Same split rules as the reference's gilbert_xyz2d_r (jenga_gilbert.py:84-288), visiting sub-cuboids in curve order.
void gen(Emitter& e, V3 p, V3 a, V3 b, V3 c) {
    const int w = len(a), h = len(b), d = len(c);
    const V3 da = unit(a), db = unit(b), dc = unit(c);
    if (h == 1 && d == 1) { for (int i = 0; i < w; ++i) { e.put(p); p = p + da; } return; }
    if (w == 1 && d == 1) { for (int i = 0; i < h; ++i) { e.put(p); p = p + db; } return; }
    if (w == 1 && h == 1) { for (int i = 0; i < d; ++i) { e.put(p); p = p + dc; } return; }
    V3 a2 = half(a), b2 = half(b), c2 = half(c);
    const int w2 = len(a2), h2 = len(b2), d2 = len(c2);
    if ((w2 % 2) && w > 2) a2 = a2 + da;  // prefer even steps
    if ((h2 % 2) && h > 2) b2 = b2 + db;
    if ((d2 % 2) && d > 2) c2 = c2 + dc;
    if (2 * w > 3 * h && 2 * w > 3 * d) {            // wide: split in w only
        gen(e, p, a2, b, c);
        gen(e, p + a2, a - a2, b, c);
    } else if (3 * h > 4 * d) {                      // do not split in d
        gen(e, p, b2, c, a2);
        gen(e, p + b2, a, b - b2, c);
        gen(e, p + (a - da) + (b2 - db), -b2, c, -(a - a2));
    } else if (3 * d > 4 * h) {                      // do not split in h
        gen(e, p, c2, a2, b);
        gen(e, p + c2, a, b, c - c2);
        gen(e, p + (a - da) + (c2 - dc), -c2, -(a - a2), b);
    } else {                                         // regular: split in all three
        gen(e, p, b2, c2, a2);
        gen(e, p + b2, c, a2, b - b2);
        gen(e, p + (b2 - db) + (c - dc), a, -b2, -(c - c2));
        gen(e, p + (a - da) + b2 + (c - dc), -c, -(a - a2), b - b2);
        gen(e, p + (a - da) + (b2 - db), -b2, c2, -(a - a2));
    }
}

// axis vectors for the (w, h, t) box: x spans w, y spans h, z spans t (reference gilbert_xyz2d :12-54)
int setup_axes(int t, int h, int w, const char* axis_order, V3 out[3]) {
    if (axis_order && axis_order[0]) {
        for (int i = 0; i < 3; ++i) {
            switch (axis_order[i]) {
                case 'w': out[i] = {w, 0, 0}; break;
                case 'h': out[i] = {0, h, 0}; break;
                case 't': out[i] = {0, 0, t}; break;
                default: return RSA_ERR_BAD_ARG;
            }
        }
        if (axis_order[0] == axis_order[1] || axis_order[0] == axis_order[2] || axis_order[1] == axis_order[2])
            return RSA_ERR_BAD_ARG;
        return RSA_OK;
    }
    if (w >= h && w >= t) { out[0] = {w, 0, 0}; out[1] = {0, h, 0}; out[2] = {0, 0, t}; }
    else if (h >= w && h >= t) { out[0] = {0, h, 0}; out[1] = {w, 0, 0}; out[2] = {0, 0, t}; }
    else { out[0] = {0, 0, t}; out[1] = {w, 0, 0}; out[2] = {0, h, 0}; }
    return RSA_OK;
}

}  // namespace

extern "C" int rsa_gilbert_mapping(int t, int h, int w, const char* axis_order, int32_t* linear_to_hilbert,
                                   int32_t* hilbert_to_linear) {
    if (t <= 0 || h <= 0 || w <= 0 || !hilbert_to_linear) return RSA_ERR_BAD_ARG;
    if ((long)t * h * w > 0x7FFFFFFFL) return RSA_ERR_UNSUPPORTED;
    V3 ax[3];
    int st = setup_axes(t, h, w, axis_order, ax);
    if (st != RSA_OK) return st;
    Emitter e{hilbert_to_linear, 0, h, w};
    gen(e, {0, 0, 0}, ax[0], ax[1], ax[2]);
    const long n = (long)t * h * w;
    if (e.n != n) return RSA_ERR_BAD_ARG;  // cannot happen for a valid cuboid
    if (linear_to_hilbert)
        for (long i = 0; i < n; ++i) linear_to_hilbert[hilbert_to_linear[i]] = (int32_t)i;
    return RSA_OK;
}

extern "C" int rsa_gilbert_block_neighbors(int t, int h, int w, int block_size, const char* axis_order,
                                           uint8_t* neighbor) {
    if (t <= 0 || h <= 0 || w <= 0 || block_size <= 0 || !neighbor) return RSA_ERR_BAD_ARG;
    const long n = (long)t * h * w;
    if (n > 0x7FFFFFFFL) return RSA_ERR_UNSUPPORTED;
    const long nb = (n + block_size - 1) / block_size;
    std::vector<int32_t> order(n), blk(n);
    int st = rsa_gilbert_mapping(t, h, w, axis_order, nullptr, order.data());
    if (st != RSA_OK) return st;
    for (long i = 0; i < n; ++i) blk[order[i]] = (int32_t)(i / block_size);  // linear index -> block of its curve index
    memset(neighbor, 0, (size_t)nb * nb);
    for (int z = 0; z < t; ++z)
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int32_t cur = blk[((long)z * h + y) * w + x];
                uint8_t* row = neighbor + (long)cur * nb;
                for (int dz = -1; dz <= 1; ++dz) {
                    const int nz = z + dz;
                    if (nz < 0 || nz >= t) continue;
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int ny = y + dy;
                        if (ny < 0 || ny >= h) continue;
                        for (int dx = -1; dx <= 1; ++dx) {
                            const int nx = x + dx;
                            if (nx < 0 || nx >= w) continue;
                            row[blk[((long)nz * h + ny) * w + nx]] = 1;  // includes the block itself
                        }
                    }
                }
            }
    return RSA_OK;
}
