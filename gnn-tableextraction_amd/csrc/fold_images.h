// P3 images of parameter sub-matrices written by the kernel that updates the parameters (the fold + Adam launch of the one-GPU
// step, gte_core.hip; the Adam launch behind the gradient all-reduce of the data-parallel step, loss_optim.hip): the operand images
// the planes GEMMs of the NEXT step multiply (csrc/p3.h).  No counterpart in the reference (torch.optim.Adam, model_train.py:168).
#pragma once
#include "gte_common.h"
#include "p3.h"

namespace gte {
// The thread that updates a parameter element writes its three bf16
// pieces into every image that holds it.  off = element offset of the sub-matrix in the flat parameter buffer; its rows x cols
// elements sit at row stride ld; image(r, c) = transpose ? sub(c, r) : sub(r, c); block fb of image row r at r * ldp + fb * bsp.
constexpr int kMaxFoldImages = 12;
struct FoldImage { long long off; unsigned span, ld, magic; int cols, transpose, pad; char* dst; long long ldp, bsp; };
struct FoldImages { FoldImage im[kMaxFoldImages]; int n; };

// (row, column) of element o of a sub-matrix with row stride ld: magic = floor(2^32 / ld) + 1 is exact for o < 2^32 / ld
// (checked on the host); a 64-bit division per element cost the launch more than the conversion launch it replaces
__device__ __forceinline__ void fold_rc(const FoldImage& im, unsigned o, int& r, int& c) {
    unsigned q = __umulhi(o, im.magic);
    unsigned rem = o - q * im.ld;
    if (rem >= im.ld) { rem -= im.ld; ++q; }
    r = (int)q; c = (int)rem;
}
__device__ __forceinline__ void fold_write_image1(const FoldImage& im, long long i, float v) {
    const long long o = i - im.off;
    if (o < 0 || o >= (long long)im.span) return;
    int r, c;
    fold_rc(im, (unsigned)o, r, c);
    if (c >= im.cols) return;
    const int ir = im.transpose ? c : r, ic = im.transpose ? r : c;
    unsigned h, m, l;
    p3::split2(v, 0.f, h, m, l);
    unsigned short* q = reinterpret_cast<unsigned short*>(im.dst + (long long)ir * im.ldp + (long long)(ic >> 4) * im.bsp + (ic & 15) * 2);
    q[0] = (unsigned short)h;
    q[p3::PLANE_BYTES / 2] = (unsigned short)m;
    q[p3::PLANE_BYTES] = (unsigned short)l;
}
// the images in `mask` only (FoldDesc::img_mask: the images whose source overlaps the fold's destination)
__device__ __forceinline__ void fold_write_images(const FoldImages& fi, unsigned mask, long long i, float v) {
    for (int k = 0; k < fi.n; ++k)
        if (mask >> k & 1) fold_write_image1(fi.im[k], i, v);
}
// two consecutive parameter elements: where both fall into the same row of an untransposed image at an even column they
// leave as ONE 4-byte store per plane; everything else goes element by element
__device__ __forceinline__ void fold_write_images2(const FoldImages& fi, unsigned mask, long long i, float v0, float v1) {
    for (int k = 0; k < fi.n; ++k) {
        if (!(mask >> k & 1)) continue;
        const FoldImage& im = fi.im[k];
        const long long o = i - im.off;
        if (o + 1 < 0 || o >= (long long)im.span) continue;
        if (o >= 0 && !im.transpose) {
            int r, c;
            fold_rc(im, (unsigned)o, r, c);
            if ((c & 1) == 0 && c + 1 < im.cols) {
                unsigned h, m, l;
                p3::split2(v0, v1, h, m, l);
                unsigned* q = reinterpret_cast<unsigned*>(im.dst + (long long)r * im.ldp + (long long)(c >> 4) * im.bsp + (c & 15) * 2);
                q[0] = h;
                q[p3::PLANE_BYTES / 4] = m;
                q[p3::PLANE_BYTES / 2] = l;
                continue;
            }
        }
        fold_write_image1(im, i, v0);
        fold_write_image1(im, i + 1, v1);
    }
}

// host side: gte_p3_desc[] (sub-matrices of the flat parameter buffer `param` [n]) -> FoldImages.  Returns GTE_OK and fi.n = n_images,
// fi.n = 0 when the list is valid but cannot ride in the launch (too many images, a row split that is not exact), or an error code
inline int make_fold_images(const float* param, int64_t n, const gte_p3_desc* images, int n_images, FoldImages& fi, const char* who) {
    fi.n = 0;
    if (n_images < 0 || (n_images > 0 && !images)) return gte::fail(GTE_ERR_INVALID_ARGUMENT, "%s: bad image list", who);
    bool images_ok = n_images > 0 && n_images <= kMaxFoldImages;
    for (int k = 0; images_ok && k < n_images; ++k) {
        const gte_p3_desc& d = images[k];
        const int64_t srows = d.transpose ? d.cols : d.rows, scols = d.transpose ? d.rows : d.cols;
        // (a negative ldp: a block-major image, block stride -ldp, rows at 96 bytes -- include/gte.h)
        const bool ldp_ok = d.ldp < 0 ? (-d.ldp >= d.rows * 96 && (-d.ldp) % 16 == 0) : d.ldp >= p3::row_bytes(d.cols);
        if (!d.src || !d.dst || d.rows <= 0 || d.cols <= 0 || d.ld < scols || d.ld > INT32_MAX || !ldp_ok ||
            d.src < param || (d.src - param) + (srows - 1) * d.ld + scols > n)
            return gte::fail(GTE_ERR_INVALID_ARGUMENT, "%s: image %d does not describe a sub-matrix of the parameters", who, k);
        FoldImage& im = fi.im[k];
        const int64_t span = (srows - 1) * d.ld + scols;
        // (the multiply-high row split would not be exact; ld == 1: the magic constant 2^32 + 1 does not fit 32 bits)
        if (d.ld < 2 || span >= ((int64_t)1 << 32) / d.ld) { images_ok = false; break; }
        im.off = d.src - param; im.span = (unsigned)span; im.ld = (unsigned)d.ld; im.magic = (unsigned)((((uint64_t)1 << 32) / d.ld) + 1);
        im.cols = (int)scols; im.transpose = d.transpose ? 1 : 0; im.pad = 0; im.dst = reinterpret_cast<char*>(d.dst);
        im.ldp = d.ldp < 0 ? 96 : d.ldp; im.bsp = d.ldp < 0 ? -d.ldp : p3::BLOCK_BYTES;
    }
    if (images_ok) fi.n = n_images;
    return GTE_OK;
}
}  // namespace gte
