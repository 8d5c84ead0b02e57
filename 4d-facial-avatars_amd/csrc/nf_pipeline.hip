// nf_render_rays_fwd: the whole per-chunk inference pipeline of predict_and_render_radiance (reference
// nerf/train_utils.py:36-162) behind ONE C entry point -- coarse depths, coarse MLP, integrator, hierarchical resampling,
// fine MLP, integrator, 7-tuple -- for callers that are not the Python package ("a fused nf_render_rays_fwd
// (rays -> 7-tuple) used by eval").  It only sequences the kernels of this library on the caller's stream; all
// intermediates live in a caller-provided workspace.
#include "nf_common.h"
#include "nf_mlp_layout.h"

__global__ void __launch_bounds__(256) k_last_column(const float* __restrict__ w, int64_t n_rows, int n_cols, float* __restrict__ out) {
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
        out[r] = w[r * n_cols + n_cols - 1];
}

static size_t nf_align64(size_t n) { return (n + 63) & ~(size_t)63; }

extern "C" size_t nf_render_rays_workspace_floats(int64_t n_rays, int n_coarse, int n_fine) {
    const size_t R = (size_t)n_rays, nc = (size_t)n_coarse, nt = (size_t)(n_coarse + n_fine);
    return 2 * nf_align64(nf_paper_cond_floats()) + nf_align64(R * nc) * 2 + nf_align64(R * nc * 4) + nf_align64(R * nt) * 2 +
           nf_align64(R * nt * 4);
}

// split_kind: what the packed_split_* streams are -- 0: (hi, lo) bf16 streams (nf_paper_pack_bf16), 1: fp16 streams (nf_paper_pack_f16),
// 2: the same fp16 streams evaluated with two products per weight ("f16x2", nf_mlp_f16x2.hip)
static int nf_render_rays_impl(int split_kind, const float* packed_coarse, const void* packed_bf16_coarse, const float* packed_fine,
                                  const void* packed_bf16_fine, const float* expr76, const float* latent32, const float* ro,
                                  const float* rd, const float* rd_view, const float* bg, const float* t_vals, const float* t_rand,
                                  const float* u, int64_t u_row_stride, const float* noise_coarse, const float* noise_fine,
                                  int64_t n_rays, int n_coarse, int n_fine, float near_z, float far_z, int white_background,
                                  float* workspace, size_t workspace_floats, float* rgb_coarse, float* disp_coarse, float* acc_coarse,
                                  float* rgb_fine, float* disp_fine, float* acc_fine, float* w_last, nf_stream_t stream) {
    if (!packed_coarse || !expr76 || !latent32 || !ro || !rd || !t_vals || !workspace || !rgb_coarse || !disp_coarse || !acc_coarse ||
        !w_last || n_rays < 0 || n_coarse <= 0 || n_fine < 0)
        return NF_EINVAL;
    const bool fine = n_fine > 0;
    if (fine && (!packed_fine || !u || !rgb_fine || !disp_fine || !acc_fine)) return NF_EINVAL;
    if (workspace_floats < nf_render_rays_workspace_floats(n_rays, n_coarse, n_fine)) return NF_EINVAL;
    if (n_rays == 0) return 0;
    const size_t R = (size_t)n_rays, nc = (size_t)n_coarse, nt = (size_t)(n_coarse + n_fine);
    float* p = workspace;
    float* cond_c = p; p += nf_align64(nf_paper_cond_floats());
    float* cond_f = p; p += nf_align64(nf_paper_cond_floats());
    float* z_c = p; p += nf_align64(R * nc);
    float* w_c = p; p += nf_align64(R * nc);
    float* raw_c = p; p += nf_align64(R * nc * 4);
    float* z_f = p; p += nf_align64(R * nt);
    float* w_f = p; p += nf_align64(R * nt);
    float* raw_f = p;
    int rc;
#define NF_TRY(call) do { rc = (call); if (rc) return rc; } while (0)
    NF_TRY(nf_paper_condition(packed_coarse, expr76, latent32, near_z, far_z, cond_c, stream));
    NF_TRY(nf_sample_coarse(n_rays, n_coarse, near_z, far_z, t_vals, t_rand, z_c, stream));
    auto split_fwd = split_kind == 2 ? nf_paper_mlp_fwd_f16x2 : (split_kind ? nf_paper_mlp_fwd_f16 : nf_paper_mlp_fwd_bf16);
    if (packed_bf16_coarse) NF_TRY(split_fwd(packed_bf16_coarse, cond_c, ro, rd, rd_view, z_c, n_rays, n_coarse, raw_c, stream));
    else NF_TRY(nf_paper_mlp_fwd(packed_coarse, cond_c, ro, rd, rd_view, z_c, n_rays, n_coarse, raw_c, stream));
    NF_TRY(nf_volume_render_fwd(raw_c, z_c, rd, noise_coarse, bg, n_rays, n_coarse, white_background, rgb_coarse, disp_coarse, acc_coarse,
                                w_c, stream));
    const float* w_src = w_c;
    int n_last = n_coarse;
    if (fine) {
        NF_TRY(nf_resample_merge(z_c, w_c, u, u_row_stride, n_rays, n_coarse, n_fine, nullptr, z_f, stream));
        NF_TRY(nf_paper_condition(packed_fine, expr76, latent32, near_z, far_z, cond_f, stream));
        if (packed_bf16_fine) NF_TRY(split_fwd(packed_bf16_fine, cond_f, ro, rd, rd_view, z_f, n_rays, (int)nt, raw_f, stream));
        else NF_TRY(nf_paper_mlp_fwd(packed_fine, cond_f, ro, rd, rd_view, z_f, n_rays, (int)nt, raw_f, stream));
        NF_TRY(nf_volume_render_fwd(raw_f, z_f, rd, noise_fine, bg, n_rays, (int)nt, white_background, rgb_fine, disp_fine, acc_fine, w_f,
                                    stream));
        w_src = w_f;
        n_last = (int)nt;
    }
#undef NF_TRY
    const int grid = (int)((n_rays + 255) / 256 < 2048 ? (n_rays + 255) / 256 : 2048);
    hipLaunchKernelGGL(k_last_column, dim3(grid), dim3(256), 0, nf_s(stream), w_src, n_rays, n_last, w_last);
    NF_RETURN_LAUNCH();
}

extern "C" int nf_render_rays_fwd(const float* packed_coarse, const void* packed_bf16_coarse, const float* packed_fine,
                                  const void* packed_bf16_fine, const float* expr76, const float* latent32, const float* ro,
                                  const float* rd, const float* rd_view, const float* bg, const float* t_vals, const float* t_rand,
                                  const float* u, int64_t u_row_stride, const float* noise_coarse, const float* noise_fine,
                                  int64_t n_rays, int n_coarse, int n_fine, float near_z, float far_z, int white_background,
                                  float* workspace, size_t workspace_floats, float* rgb_coarse, float* disp_coarse, float* acc_coarse,
                                  float* rgb_fine, float* disp_fine, float* acc_fine, float* w_last, nf_stream_t stream) {
    return nf_render_rays_impl(0, packed_coarse, packed_bf16_coarse, packed_fine, packed_bf16_fine, expr76, latent32, ro, rd, rd_view, bg, t_vals,
                               t_rand, u, u_row_stride, noise_coarse, noise_fine, n_rays, n_coarse, n_fine, near_z, far_z, white_background,
                               workspace, workspace_floats, rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine, w_last, stream);
}

// the same pipeline on the split-fp16 MLP kernels: packed_f16_* = streams of nf_paper_pack_f16 (NULL: that network runs exact f32)
extern "C" int nf_render_rays_fwd_f16(const float* packed_coarse, const void* packed_f16_coarse, const float* packed_fine,
                                      const void* packed_f16_fine, const float* expr76, const float* latent32, const float* ro,
                                      const float* rd, const float* rd_view, const float* bg, const float* t_vals, const float* t_rand,
                                      const float* u, int64_t u_row_stride, const float* noise_coarse, const float* noise_fine,
                                      int64_t n_rays, int n_coarse, int n_fine, float near_z, float far_z, int white_background,
                                      float* workspace, size_t workspace_floats, float* rgb_coarse, float* disp_coarse, float* acc_coarse,
                                      float* rgb_fine, float* disp_fine, float* acc_fine, float* w_last, nf_stream_t stream) {
    return nf_render_rays_impl(1, packed_coarse, packed_f16_coarse, packed_fine, packed_f16_fine, expr76, latent32, ro, rd, rd_view, bg, t_vals,
                               t_rand, u, u_row_stride, noise_coarse, noise_fine, n_rays, n_coarse, n_fine, near_z, far_z, white_background,
                               workspace, workspace_floats, rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine, w_last, stream);
}

// ... and with two fp16 products per weight ("f16x2"): same streams, same arguments
extern "C" int nf_render_rays_fwd_f16x2(const float* packed_coarse, const void* packed_f16_coarse, const float* packed_fine,
                                        const void* packed_f16_fine, const float* expr76, const float* latent32, const float* ro,
                                        const float* rd, const float* rd_view, const float* bg, const float* t_vals, const float* t_rand,
                                        const float* u, int64_t u_row_stride, const float* noise_coarse, const float* noise_fine,
                                        int64_t n_rays, int n_coarse, int n_fine, float near_z, float far_z, int white_background,
                                        float* workspace, size_t workspace_floats, float* rgb_coarse, float* disp_coarse, float* acc_coarse,
                                        float* rgb_fine, float* disp_fine, float* acc_fine, float* w_last, nf_stream_t stream) {
    return nf_render_rays_impl(2, packed_coarse, packed_f16_coarse, packed_fine, packed_f16_fine, expr76, latent32, ro, rd, rd_view, bg, t_vals,
                               t_rand, u, u_row_stride, noise_coarse, noise_fine, n_rays, n_coarse, n_fine, near_z, far_z, white_background,
                               workspace, workspace_floats, rgb_coarse, disp_coarse, acc_coarse, rgb_fine, disp_fine, acc_fine, w_last, stream);
}
