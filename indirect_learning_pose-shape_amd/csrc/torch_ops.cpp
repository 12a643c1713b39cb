// TORCH_LIBRARY(smplraster, ...): the at::Tensor layer SURVEY.md section 8(b) specifies on top of the extern "C"
// launchers of libsmplraster_hip.so ("C-ABI / extension layer beneath": smpl_fwd / smpl_bwd, project_fwd / bwd,
// visibility, seg_fwd / seg_bwd, silh_fwd / bwd - all contiguous fp32 / int tensors on one HIP device, launched on
// at::hip::getCurrentHIPStream()).  One host call per op instead of a dozen ctypes marshalling steps: outputs and
// workspaces are allocated here (at::empty on the caching allocator - no hipMalloc, so the ops stay HIP-graph
// capturable like the launchers), arguments are TORCH_CHECKed, launcher errors become c10::Error with
// smplr_last_error()'s text.  Meta kernels give the output shapes, so the ops trace under torch.compile / export.
// The ctypes table (_lib.py) stays: it is the torch-free binding of the same library and what the autograd Functions
// of ops.py use.  Reference call sites these ops stand for: model.py:108-118 (decoder wiring),
// keras_smpl/batch_smpl.py:96-153, projection.py:54-81, compute_mask.py:12-108, projects_to_seg.py:9-69,
// projects_to_silhouette.py:14-44.
//
// No compute happens here and nothing falls back: a CPU tensor is refused, a launcher failure raises.
#include <ATen/ATen.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>
#include <torch/library.h>

#include <initializer_list>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/smplraster.h"

namespace {

using at::Tensor;

// (torch on ROCm keeps its device type named "cuda": the guard and the stream are the "masquerading" forms - the plain
// c10::hip ones belong to a HIP device type torch tensors never carry, and their current stream is not torch's)
using DeviceGuard = c10::hip::HIPGuardMasqueradingAsCUDA;
void *cur_stream() { return reinterpret_cast<void *>(c10::hip::getCurrentHIPStreamMasqueradingAsCUDA().stream()); }

void ok(int rc, const char *what) {
  TORCH_CHECK(rc == 0, what, " failed (rc=", rc, "): ", smplr_last_error());
}

const Tensor &dev_f32(const Tensor &t, const char *name) {
  TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (got ", t.device(), "); this library has no CPU path");
  TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  return t;
}
const Tensor &dev_typed(const Tensor &t, at::ScalarType ty, const char *name) {
  TORCH_CHECK(t.is_cuda(), name, " must live on a HIP device (got ", t.device(), ")");
  TORCH_CHECK(t.scalar_type() == ty, name, " has the wrong dtype");
  TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
  return t;
}
// every operand of a launch on the device the guard selects (a CPU constant or a tensor of another GPU would be a page
// fault inside the kernel, not an error)
void same_device(const Tensor &ref, std::initializer_list<std::pair<const char *, const Tensor *>> ts) {
  for (const auto &nt : ts) {
    if (!nt.second->defined() || nt.second->numel() == 0) continue;
    TORCH_CHECK(nt.second->device() == ref.device(), nt.first, " lives on ", nt.second->device(), ", the launch runs on ", ref.device());
  }
}
const float *fp(const Tensor &t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }
float *fpm(Tensor &t) { return t.defined() && t.numel() ? t.data_ptr<float>() : nullptr; }
Tensor bytes(size_t n, const Tensor &like) {
  return at::empty({(int64_t)(n < 16 ? 16 : n)}, like.options().dtype(at::kByte));
}
Tensor f32(at::IntArrayRef shape, const Tensor &like) { return at::empty(shape, like.options().dtype(at::kFloat)); }

// ---- compute_mask ------------------------------------------------------------------------------------------
Tensor visibility(const Tensor &proj, int64_t grid_wh, bool ref_compat) {
  dev_f32(proj, "proj");
  TORCH_CHECK(proj.dim() == 3 && proj.size(2) == 3, "proj must be (B, VP, 3)");
  DeviceGuard g(proj.device());
  Tensor mask = f32({proj.size(0), proj.size(1)}, proj);
  ok(smplr_visibility(fp(proj), (int)proj.size(0), (int)proj.size(1), (int)grid_wh, ref_compat ? 1 : 0, fpm(mask),
                      cur_stream()), "smplr_visibility");
  return mask;
}
Tensor visibility_meta(const Tensor &proj, int64_t, bool) { return at::empty({proj.size(0), proj.size(1)}, proj.options()); }

// ---- orthographic_project ---------------------------------------------------------------------------------
Tensor project_fwd(const Tensor &verts, const Tensor &cam, int64_t vs) {
  dev_f32(verts, "verts");
  dev_f32(cam, "cam");
  TORCH_CHECK(verts.dim() == 3 && verts.size(2) == 3 && cam.dim() == 2 && cam.size(0) == verts.size(0) && cam.size(1) >= 4 && vs >= 1,
              "project_fwd: verts (B,V,3), cam (B,>=4), vertex_sampling >= 1");
  same_device(verts, {{"cam", &cam}});
  DeviceGuard g(verts.device());
  const int64_t B = verts.size(0), V = verts.size(1), VP = (V + vs - 1) / vs;
  Tensor proj = f32({B, VP, 3}, verts);
  ok(smplr_project_fwd(fp(verts), fp(cam), (int)cam.size(1), (int)B, (int)V, (int)vs, fpm(proj), cur_stream()),
     "smplr_project_fwd");
  return proj;
}
Tensor project_fwd_meta(const Tensor &verts, const Tensor &, int64_t vs) {
  return at::empty({verts.size(0), (verts.size(1) + vs - 1) / vs, 3}, verts.options());
}
std::tuple<Tensor, Tensor> project_bwd(const Tensor &dproj, const Tensor &verts, const Tensor &cam, int64_t vs) {
  dev_f32(dproj, "dproj");
  dev_f32(verts, "verts");
  dev_f32(cam, "cam");
  TORCH_CHECK(verts.dim() == 3 && verts.size(2) == 3 && cam.dim() == 2 && cam.size(0) == verts.size(0) && cam.size(1) >= 4 && vs >= 1,
              "project_bwd: verts (B,V,3), cam (B,>=4), vertex_sampling >= 1");
  same_device(verts, {{"dproj", &dproj}, {"cam", &cam}});
  DeviceGuard g(verts.device());
  const int64_t B = verts.size(0), V = verts.size(1);
  TORCH_CHECK(dproj.dim() == 3 && dproj.size(0) == B && dproj.size(1) == (V + vs - 1) / vs && dproj.size(2) == 3, "dproj must be (B, VP, 3)");
  Tensor dverts = f32({B, V, 3}, verts), dcam = f32({B, 4}, verts);
  ok(smplr_project_bwd(fp(dproj), fp(verts), fp(cam), (int)cam.size(1), (int)B, (int)V, (int)vs, fpm(dverts), fpm(dcam),
                       cur_stream()), "smplr_project_bwd");
  return {dverts, dcam};
}
std::tuple<Tensor, Tensor> project_bwd_meta(const Tensor &, const Tensor &verts, const Tensor &, int64_t) {
  return {at::empty_like(verts), at::empty({verts.size(0), 4}, verts.options())};
}

// ---- projects_to_seg --------------------------------------------------------------------------------------
struct PartDims { int P, K; };
PartDims part_dims(const Tensor &part_pos, const Tensor &part_off) {
  dev_typed(part_pos, at::kInt, "part_pos");
  dev_typed(part_off, at::kInt, "part_off");
  TORCH_CHECK(part_off.numel() >= 2 && part_pos.numel() >= 1, "empty part table");
  return {(int)part_off.numel() - 1, (int)part_pos.numel()};
}
// -> seg (B,W,W,P+1), arg (B,W,W,32) int16, rec (B,S,4)
std::tuple<Tensor, Tensor, Tensor> seg_fwd(const Tensor &proj, const Tensor &mask, const Tensor &part_pos,
                                           const Tensor &part_off, int64_t W) {
  dev_f32(proj, "proj");
  dev_f32(mask, "mask");
  const PartDims d = part_dims(part_pos, part_off);
  TORCH_CHECK(proj.dim() == 3 && proj.size(2) == 3 && mask.dim() == 2 && mask.size(0) == proj.size(0) && mask.size(1) == proj.size(1),
              "seg_fwd: proj (B,VP,3), mask (B,VP)");
  TORCH_CHECK(W > 0 && W <= 160, "seg_fwd: 0 < W <= 160");
  same_device(proj, {{"mask", &mask}, {"part_pos", &part_pos}, {"part_off", &part_off}});
  DeviceGuard g(proj.device());
  const int64_t B = proj.size(0), VP = proj.size(1);
  Tensor seg = f32({B, W, W, d.P + 1}, proj), arg = at::empty({B, W, W, 32}, proj.options().dtype(at::kShort));
  Tensor rec = f32({B, smplr_seg_slots(d.P, d.K), 4}, proj);
  Tensor ws = bytes(smplr_seg_workspace((int)B, (int)VP, (int)W, d.P, d.K), proj);
  ok(smplr_seg_fwd(fp(proj), fp(mask), (int)B, (int)VP, (int)W, part_pos.data_ptr<int32_t>(), part_off.data_ptr<int32_t>(),
                   d.P, d.K, ws.data_ptr(), fpm(seg), arg.data_ptr<int16_t>(), fpm(rec), nullptr, cur_stream()),
     "smplr_seg_fwd");
  return {seg, arg, rec};
}
std::tuple<Tensor, Tensor, Tensor> seg_fwd_meta(const Tensor &proj, const Tensor &, const Tensor &part_pos,
                                                const Tensor &part_off, int64_t W) {
  const int P = (int)part_off.numel() - 1, K = (int)part_pos.numel();
  return {at::empty({proj.size(0), W, W, P + 1}, proj.options()), at::empty({proj.size(0), W, W, 32}, proj.options().dtype(at::kShort)),
          at::empty({proj.size(0), smplr_seg_slots(P, K), 4}, proj.options())};
}
Tensor seg_bwd(const Tensor &dseg, const Tensor &arg, const Tensor &rec, int64_t VP, int64_t P, int64_t K, bool deterministic) {
  dev_f32(dseg, "dseg");
  dev_typed(arg, at::kShort, "arg");
  dev_f32(rec, "rec");
  TORCH_CHECK(dseg.dim() == 4 && dseg.size(1) == dseg.size(2) && dseg.size(3) == P + 1, "dseg must be (B,W,W,P+1)");
  const int64_t B = dseg.size(0), W = dseg.size(1);
  TORCH_CHECK(P >= 1 && P <= 31 && K >= 1 && VP >= 1, "seg_bwd: 1 <= P <= 31, K >= 1, VP >= 1");
  TORCH_CHECK(arg.dim() == 4 && arg.size(0) == B && arg.size(1) == W && arg.size(2) == W && arg.size(3) == 32,
              "arg must be (B,W,W,32) int16 as seg_fwd returned it");
  TORCH_CHECK(rec.dim() == 3 && rec.size(0) == B && rec.size(1) == smplr_seg_slots((int)P, (int)K) && rec.size(2) == 4,
              "rec must be (B, smplr_seg_slots(P,K), 4) as seg_fwd returned it for this part table");
  same_device(dseg, {{"arg", &arg}, {"rec", &rec}});
  DeviceGuard g(dseg.device());
  Tensor dproj = f32({B, VP, 3}, dseg);
  Tensor ws = bytes(smplr_seg_bwd_workspace((int)B, (int)W), dseg);
  ok(smplr_seg_bwd(fp(dseg), arg.data_ptr<int16_t>(), fp(rec), (int)B, (int)VP, (int)W, (int)P, (int)K, fpm(dproj),
                   ws.data_ptr(), deterministic ? 1 : 0, cur_stream()), "smplr_seg_bwd");
  return dproj;
}
Tensor seg_bwd_meta(const Tensor &dseg, const Tensor &, const Tensor &, int64_t VP, int64_t, int64_t, bool) {
  return at::empty({dseg.size(0), VP, 3}, dseg.options());
}

// ---- projects_to_silhouette -------------------------------------------------------------------------------
std::tuple<Tensor, Tensor> silh_fwd(const Tensor &proj, int64_t W) {
  dev_f32(proj, "proj");
  TORCH_CHECK(proj.dim() == 3 && proj.size(2) == 3, "proj must be (B, VP, 3)");
  TORCH_CHECK(W > 0, "silh_fwd: W > 0");
  DeviceGuard g(proj.device());
  const int64_t B = proj.size(0), VP = proj.size(1);
  Tensor silh = f32({B, W, W, 2}, proj), sarg = at::empty({B, W, W}, proj.options().dtype(at::kInt));
  Tensor ws = bytes(smplr_silh_workspace((int)B, (int)VP, (int)W), proj);
  ok(smplr_silh_fwd(fp(proj), (int)B, (int)VP, (int)W, fpm(silh), sarg.data_ptr<int32_t>(), ws.data_ptr(), cur_stream()),
     "smplr_silh_fwd");
  return {silh, sarg};
}
std::tuple<Tensor, Tensor> silh_fwd_meta(const Tensor &proj, int64_t W) {
  return {at::empty({proj.size(0), W, W, 2}, proj.options()), at::empty({proj.size(0), W, W}, proj.options().dtype(at::kInt))};
}
Tensor silh_bwd(const Tensor &dsilh, const Tensor &silh, const Tensor &sarg, const Tensor &proj, bool deterministic) {
  dev_f32(dsilh, "dsilh");
  dev_f32(silh, "silh");
  dev_typed(sarg, at::kInt, "arg");
  dev_f32(proj, "proj");
  TORCH_CHECK(proj.dim() == 3 && proj.size(2) == 3, "proj must be (B, VP, 3)");
  TORCH_CHECK(silh.dim() == 4 && silh.size(0) == proj.size(0) && silh.size(1) == silh.size(2) && silh.size(3) == 2,
              "silh must be (B,W,W,2) as silh_fwd returned it");
  TORCH_CHECK(dsilh.sizes() == silh.sizes(), "dsilh must have silh's shape");
  TORCH_CHECK(sarg.dim() == 3 && sarg.size(0) == silh.size(0) && sarg.size(1) == silh.size(1) && sarg.size(2) == silh.size(2),
              "arg must be (B,W,W) int32 as silh_fwd returned it");
  same_device(proj, {{"dsilh", &dsilh}, {"silh", &silh}, {"arg", &sarg}});
  DeviceGuard g(proj.device());
  const int64_t B = proj.size(0), VP = proj.size(1), W = silh.size(1);
  Tensor dproj = f32({B, VP, 3}, proj);
  ok(smplr_silh_bwd(fp(dsilh), fp(silh), sarg.data_ptr<int32_t>(), fp(proj), (int)B, (int)VP, (int)W, fpm(dproj),
                    deterministic ? 1 : 0, cur_stream()), "smplr_silh_bwd");
  return dproj;
}
Tensor silh_bwd_meta(const Tensor &, const Tensor &, const Tensor &, const Tensor &proj, bool) { return at::empty_like(proj); }

// ---- SMPLLayer.call ---------------------------------------------------------------------------------------
// consts (the device constants of SMPLLayer.build, ops.SMPLConstants.as_list()):
//   [0] J_template (24,3)  [1] J_dirs (24,3,10)  [2] parents (24) int32  [3] v_template (3V)
//   [4] blend3_fwd (bytes: smplr_blend3_pack's forward operand)  [5] blend3_bwd (bytes)
//   [6] lbs_weights (V,24)  [7] lbs_top4 (V,8) or an empty tensor  [8] blend_t (3V,224) (fp32 backward operand; may be empty)
struct Consts {
  const Tensor &Jt, &Jd, &par, &vt, &b3f, &b3b, &lw, &l4, &bt;
  int V;
};
Consts unpack(const std::vector<Tensor> &c) {
  TORCH_CHECK(c.size() == 9, "consts must hold 9 tensors (see ops.SMPLConstants.as_list)");
  dev_f32(c[0], "J_template");
  dev_f32(c[1], "J_dirs");
  dev_typed(c[2], at::kInt, "parents");
  dev_f32(c[3], "v_template");
  TORCH_CHECK(c[4].defined() && c[4].numel() > 0 && c[4].is_cuda(), "blend3_fwd missing: upload the constants with SMPLR_BLEND_GEMM=bf16x3");
  dev_f32(c[6], "lbs_weights");
  TORCH_CHECK(c[3].numel() % 3 == 0, "v_template must hold 3V floats");
  const int64_t V_ = c[3].numel() / 3;
  TORCH_CHECK(c[0].numel() == 72 && c[1].numel() == 720 && c[2].numel() == 24, "J_template (24,3), J_dirs (24,3,10), parents (24)");
  TORCH_CHECK(c[6].numel() == V_ * 24, "lbs_weights must be (V,24)");
  TORCH_CHECK(c[4].is_contiguous() && (size_t)c[4].nbytes() >= smplr_blend3_fwd_bytes((int)(3 * V_)),
              "blend3_fwd is smaller than smplr_blend3_fwd_bytes(3V)");
  if (c[5].defined() && c[5].numel() > 0)
    TORCH_CHECK(c[5].is_cuda() && c[5].is_contiguous() && (size_t)c[5].nbytes() >= smplr_blend3_bwd_bytes((int)(3 * V_)),
                "blend3_bwd must be a contiguous device buffer of smplr_blend3_bwd_bytes(3V)");
  if (c[7].defined() && c[7].numel() > 0) {
    dev_f32(c[7], "lbs_top4");
    TORCH_CHECK(c[7].numel() == V_ * 8, "lbs_top4 must be (V,8)");
  }
  if (c[8].defined() && c[8].numel() > 0) {
    dev_f32(c[8], "blend_t");
    TORCH_CHECK(c[8].numel() == 3 * V_ * 224, "blend_t must be (3V,224)");
  }
  same_device(c[3], {{"J_template", &c[0]}, {"J_dirs", &c[1]}, {"parents", &c[2]}, {"blend3_fwd", &c[4]}, {"blend3_bwd", &c[5]},
                     {"lbs_weights", &c[6]}, {"lbs_top4", &c[7]}, {"blend_t", &c[8]}});
  return Consts{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], (int)(c[3].numel() / 3)};
}
// -> verts (B,V,3), v_posed (B,V,3), A (B,24,12), Rs (B,24,9), J (B,24,3), J_transformed (B,24,3)
std::vector<Tensor> smpl_fwd(const Tensor &x, const std::vector<Tensor> &consts, int64_t num_cam) {
  dev_f32(x, "x");
  const Consts c = unpack(consts);
  TORCH_CHECK(x.dim() == 2 && x.size(1) == num_cam + 82, "x must be (B, num_cam + 82)");
  same_device(x, {{"consts", &c.vt}});
  DeviceGuard g(x.device());
  const int64_t B = x.size(0);
  Tensor Rs = f32({B, 24, 9}, x), J = f32({B, 24, 3}, x), A = f32({B, 24, 12}, x), Jt = f32({B, 24, 3}, x);
  Tensor v_posed = f32({B, c.V, 3}, x), verts = f32({B, c.V, 3}, x);
  ok(smplr_pose_blend3_fwd(fp(x), (int)x.size(1), (int)num_cam, (int)B, fp(c.Jt), fp(c.Jd), c.par.data_ptr<int32_t>(),
                           c.b3f.data_ptr(), fp(c.vt), 3 * c.V, fpm(Rs), fpm(J), fpm(A), fpm(Jt), fpm(v_posed), cur_stream()),
     "smplr_pose_blend3_fwd");
  ok(smplr_skin_fwd(fp(v_posed), fp(c.lw), fp(c.l4), fp(A), nullptr, 0, (int)B, c.V, 1, fpm(verts), nullptr, cur_stream()),
     "smplr_skin_fwd");
  return {verts, v_posed, A, Rs, J, Jt};
}
std::vector<Tensor> smpl_fwd_meta(const Tensor &x, const std::vector<Tensor> &consts, int64_t) {
  const int64_t B = x.size(0), V = consts.at(3).numel() / 3;
  auto o = x.options();
  return {at::empty({B, V, 3}, o), at::empty({B, V, 3}, o), at::empty({B, 24, 12}, o), at::empty({B, 24, 9}, o),
          at::empty({B, 24, 3}, o), at::empty({B, 24, 3}, o)};
}
// d x from d verts (and / or d proj, d J_transformed): smplr_smpl_bwd
Tensor smpl_bwd(const c10::optional<Tensor> &dverts, const c10::optional<Tensor> &dproj, const c10::optional<Tensor> &dJt,
                const Tensor &x, const std::vector<Tensor> &consts, const Tensor &Rs, const Tensor &J, const Tensor &A,
                const Tensor &v_posed, int64_t num_cam, int64_t vs) {
  dev_f32(x, "x");
  const Consts c = unpack(consts);
  TORCH_CHECK(dverts.has_value() || dproj.has_value(), "smpl_bwd: dverts or dproj is required");
  if (dverts) dev_f32(*dverts, "dverts");
  if (dproj) dev_f32(*dproj, "dproj");
  if (dJt) dev_f32(*dJt, "dJ_transformed");
  dev_f32(Rs, "Rs"); dev_f32(J, "J"); dev_f32(A, "A"); dev_f32(v_posed, "v_posed");
  TORCH_CHECK(c.b3b.defined() && c.b3b.numel() > 0, "blend3_bwd missing");
  TORCH_CHECK(x.dim() == 2 && x.size(1) == num_cam + 82 && vs >= 1, "x must be (B, num_cam + 82), vertex_sampling >= 1");
  const int64_t B = x.size(0);
  TORCH_CHECK(Rs.numel() == B * 216 && J.numel() == B * 72 && A.numel() == B * 288 && v_posed.numel() == B * 3 * c.V,
              "Rs (B,24,9), J (B,24,3), A (B,24,12), v_posed (B,V,3) as smpl_fwd returned them");
  if (dverts) TORCH_CHECK(dverts->numel() == B * 3 * c.V, "dverts must be (B,V,3)");
  if (dproj) TORCH_CHECK(dproj->numel() == B * 3 * ((c.V + vs - 1) / vs), "dproj must be (B,VP,3)");
  if (dJt) TORCH_CHECK(dJt->numel() == B * 72, "dJ_transformed must be (B,24,3)");
  same_device(x, {{"consts", &c.vt}, {"Rs", &Rs}, {"J", &J}, {"A", &A}, {"v_posed", &v_posed}});
  if (dverts) same_device(x, {{"dverts", &*dverts}});
  if (dproj) same_device(x, {{"dproj", &*dproj}});
  if (dJt) same_device(x, {{"dJ_transformed", &*dJt}});
  DeviceGuard g(x.device());
  Tensor dx = f32({B, x.size(1)}, x);
  Tensor ws = bytes(smplr_smpl_bwd_workspace((int)B, c.V), x);
  ok(smplr_smpl_bwd(dverts ? fp(*dverts) : nullptr, dproj ? fp(*dproj) : nullptr, nullptr, nullptr, 0, dJt ? fp(*dJt) : nullptr,
                    fp(x), (int)x.size(1), (int)num_cam, (int)B, c.V, (int)vs, fp(c.bt), c.b3b.data_ptr(), fp(c.lw), fp(c.l4),
                    fp(c.Jd), c.par.data_ptr<int32_t>(), fp(Rs), fp(J), fp(A), fp(v_posed), fpm(dx), ws.data_ptr(), cur_stream()),
     "smplr_smpl_bwd");
  return dx;
}
Tensor smpl_bwd_meta(const c10::optional<Tensor> &, const c10::optional<Tensor> &, const c10::optional<Tensor> &, const Tensor &x,
                     const std::vector<Tensor> &, const Tensor &, const Tensor &, const Tensor &, const Tensor &, int64_t, int64_t) {
  return at::empty_like(x);
}

// ---- the decoder's forward as ONE host call (model.py:108-118: SMPLLayer -> project -> compute_mask -> segment) ---
// -> verts, proj, mask, seg, J_transformed, arg, rec, vslot, v_posed, A, Rs, J   (what predict.py:112-118 asks three
// Keras models for, plus what a backward would need); the two launches of the fused path when the mesh fits the
// binning workgroup (smplr_skin_vis_seg_fits), else skinning + smplr_vis_seg_fwd.
std::vector<Tensor> decoder_fwd(const Tensor &x, const std::vector<Tensor> &consts, const Tensor &part_pos,
                                const Tensor &part_off, int64_t W, int64_t grid_wh, bool ref_compat, int64_t num_cam) {
  dev_f32(x, "x");
  const Consts c = unpack(consts);
  const PartDims d = part_dims(part_pos, part_off);
  TORCH_CHECK(x.dim() == 2 && x.size(1) == num_cam + 82, "x must be (B, num_cam + 82)");
  TORCH_CHECK(grid_wh > 0 && grid_wh <= 128 && W > 0 && W <= 160, "decoder_fwd: 0 < grid_wh <= 128, 0 < W <= 160");
  same_device(x, {{"consts", &c.vt}, {"part_pos", &part_pos}, {"part_off", &part_off}});
  DeviceGuard g(x.device());
  const int64_t B = x.size(0), V = c.V;
  Tensor Rs = f32({B, 24, 9}, x), J = f32({B, 24, 3}, x), A = f32({B, 24, 12}, x), Jt = f32({B, 24, 3}, x);
  Tensor v_posed = f32({B, V, 3}, x), verts = f32({B, V, 3}, x), proj = f32({B, V, 3}, x), mask = f32({B, V}, x);
  Tensor seg = f32({B, W, W, d.P + 1}, x), arg = at::empty({B, W, W, 32}, x.options().dtype(at::kShort));
  Tensor rec = f32({B, smplr_seg_slots(d.P, d.K), 4}, x), vslot = at::empty({B, V}, x.options().dtype(at::kShort));
  Tensor ws = bytes(smplr_seg_workspace((int)B, (int)V, (int)W, d.P, d.K), x);
  if (B > 0) {
    ok(smplr_pose_blend3_fwd(fp(x), (int)x.size(1), (int)num_cam, (int)B, fp(c.Jt), fp(c.Jd), c.par.data_ptr<int32_t>(),
                             c.b3f.data_ptr(), fp(c.vt), 3 * c.V, fpm(Rs), fpm(J), fpm(A), fpm(Jt), fpm(v_posed), cur_stream()),
       "smplr_pose_blend3_fwd");
    const bool fits = c.l4.defined() && c.l4.numel() > 0 && smplr_skin_vis_seg_fits((int)V, (int)W, (int)grid_wh) == 1;
    if (fits) {
      ok(smplr_skin_vis_seg_fwd(fp(v_posed), fp(c.l4), fp(A), fp(x), (int)x.size(1), (int)B, (int)V, (int)W, (int)grid_wh,
                                ref_compat ? 1 : 0, part_pos.data_ptr<int32_t>(), part_off.data_ptr<int32_t>(), d.P, d.K,
                                ws.data_ptr(), fpm(verts), fpm(proj), fpm(mask), fpm(seg), arg.data_ptr<int16_t>(), fpm(rec),
                                vslot.data_ptr<int16_t>(), cur_stream()), "smplr_skin_vis_seg_fwd");
    } else {
      ok(smplr_skin_fwd(fp(v_posed), fp(c.lw), fp(c.l4), fp(A), fp(x), (int)x.size(1), (int)B, (int)V, 1, fpm(verts), fpm(proj),
                        cur_stream()), "smplr_skin_fwd");
      ok(smplr_vis_seg_fwd(fp(proj), (int)B, (int)V, (int)W, (int)grid_wh, ref_compat ? 1 : 0, part_pos.data_ptr<int32_t>(),
                           part_off.data_ptr<int32_t>(), d.P, d.K, ws.data_ptr(), fpm(mask), fpm(seg), arg.data_ptr<int16_t>(),
                           fpm(rec), vslot.data_ptr<int16_t>(), cur_stream()), "smplr_vis_seg_fwd");
    }
  }
  return {verts, proj, mask, seg, Jt, arg, rec, vslot, v_posed, A, Rs, J};
}
std::vector<Tensor> decoder_fwd_meta(const Tensor &x, const std::vector<Tensor> &consts, const Tensor &part_pos,
                                     const Tensor &part_off, int64_t W, int64_t, bool, int64_t) {
  const int64_t B = x.size(0), V = consts.at(3).numel() / 3;
  const int P = (int)part_off.numel() - 1, K = (int)part_pos.numel();
  auto o = x.options();
  return {at::empty({B, V, 3}, o), at::empty({B, V, 3}, o), at::empty({B, V}, o), at::empty({B, W, W, P + 1}, o),
          at::empty({B, 24, 3}, o), at::empty({B, W, W, 32}, o.dtype(at::kShort)), at::empty({B, smplr_seg_slots(P, K), 4}, o),
          at::empty({B, V}, o.dtype(at::kShort)), at::empty({B, V, 3}, o), at::empty({B, 24, 12}, o), at::empty({B, 24, 9}, o),
          at::empty({B, 24, 3}, o)};
}

int64_t abi_version() { return smplr_abi_version(); }
#ifndef SMPLR_TORCH_OPS_ID
#define SMPLR_TORCH_OPS_ID "unknown"
#endif
// sha256 of this file (16 digits) + the torch version it was compiled against (csrc/Makefile): torch_ops.load() refuses a
// layer built from another torch_ops.cpp or for another torch
std::string build_tag() { return SMPLR_TORCH_OPS_ID; }

}  // namespace

TORCH_LIBRARY(smplraster, m) {
  m.def("abi_version() -> int", &abi_version);
  m.def("build_tag() -> str", &build_tag);
  m.def("visibility(Tensor proj, int grid_wh=64, bool ref_compat=True) -> Tensor");
  m.def("project_fwd(Tensor verts, Tensor cam, int vertex_sampling=1) -> Tensor");
  m.def("project_bwd(Tensor dproj, Tensor verts, Tensor cam, int vertex_sampling=1) -> (Tensor, Tensor)");
  m.def("seg_fwd(Tensor proj, Tensor mask, Tensor part_pos, Tensor part_off, int W) -> (Tensor, Tensor, Tensor)");
  m.def("seg_bwd(Tensor dseg, Tensor arg, Tensor rec, int VP, int P, int K, bool deterministic=False) -> Tensor");
  m.def("silh_fwd(Tensor proj, int W) -> (Tensor, Tensor)");
  m.def("silh_bwd(Tensor dsilh, Tensor silh, Tensor arg, Tensor proj, bool deterministic=False) -> Tensor");
  m.def("smpl_fwd(Tensor x, Tensor[] consts, int num_cam=4) -> Tensor[]");
  m.def("smpl_bwd(Tensor? dverts, Tensor? dproj, Tensor? dJ_transformed, Tensor x, Tensor[] consts, Tensor Rs, Tensor J, "
        "Tensor A, Tensor v_posed, int num_cam=4, int vertex_sampling=1) -> Tensor");
  m.def("decoder_fwd(Tensor x, Tensor[] consts, Tensor part_pos, Tensor part_off, int W, int grid_wh=64, "
        "bool ref_compat=True, int num_cam=4) -> Tensor[]");
}

TORCH_LIBRARY_IMPL(smplraster, CUDA, m) {       // (the HIP backend's dispatch key is named CUDA in torch)
  m.impl("visibility", &visibility);
  m.impl("project_fwd", &project_fwd);
  m.impl("project_bwd", &project_bwd);
  m.impl("seg_fwd", &seg_fwd);
  m.impl("seg_bwd", &seg_bwd);
  m.impl("silh_fwd", &silh_fwd);
  m.impl("silh_bwd", &silh_bwd);
  m.impl("smpl_fwd", &smpl_fwd);
  m.impl("smpl_bwd", &smpl_bwd);
  m.impl("decoder_fwd", &decoder_fwd);
}

TORCH_LIBRARY_IMPL(smplraster, Meta, m) {
  m.impl("visibility", &visibility_meta);
  m.impl("project_fwd", &project_fwd_meta);
  m.impl("project_bwd", &project_bwd_meta);
  m.impl("seg_fwd", &seg_fwd_meta);
  m.impl("seg_bwd", &seg_bwd_meta);
  m.impl("silh_fwd", &silh_fwd_meta);
  m.impl("silh_bwd", &silh_bwd_meta);
  m.impl("smpl_fwd", &smpl_fwd_meta);
  m.impl("smpl_bwd", &smpl_bwd_meta);
  m.impl("decoder_fwd", &decoder_fwd_meta);
}
