// Scene "compiler": validates the caller's object graph and lowers it to the threaded program of
// rl_program.h.  Host code only (no HIP calls).
#include "rl_program.h"

#include <cmath>
#include <cstring>

namespace rl {

namespace {

const uint32_t MAX_NEST = 100000;  // recursion guard (a cycle in the graph is malformed input)

struct RtiowCompiler {
  const rl_rtiow_scene_desc &d;
  RtiowProgram &p;
  std::string &err;
  std::vector<uint8_t> bvh_busy, list_busy, tr_busy, tf_busy;
  std::vector<uint32_t> push_stack;
  bool in_medium = false;

  bool fail(const std::string &m) {
    err = m;
    return false;
  }
  bool check(rl_href h) {
    switch (h.kind) {
      case RL_H_SPHERE: return h.index < d.n_spheres || fail("sphere index out of range");
      case RL_H_PLANAR: return h.index < d.n_planars || fail("planar index out of range");
      case RL_H_TRANSLATE: return h.index < d.n_translates || fail("translate index out of range");
      case RL_H_TRANSFORM: return h.index < d.n_transforms || fail("transform index out of range");
      case RL_H_BVH: return h.index < d.n_bvh_nodes || fail("bvh node index out of range");
      case RL_H_LIST: return h.index < d.n_lists || fail("list index out of range");
      case RL_H_MEDIUM: return h.index < d.n_media || fail("medium index out of range");
      default: return fail("unknown hittable kind");
    }
  }
  uint32_t sphere_payload(uint32_t idx) { return idx | (d.spheres[idx].moving ? SPH_MOVING : 0u) | (p.sphere_uv[idx] ? SPH_UV : 0u); }

  bool emit(rl_href h, uint32_t depth) {
    if (depth > MAX_NEST) return fail("scene graph too deep");
    if (!check(h)) return false;
    switch (h.kind) {
      case RL_H_SPHERE: {
        DevOp op{};
        op.code = OP_SPHERE, op.a = sphere_payload(h.index), op.b = NONE;
        op.skip = (uint32_t)p.ops.size() + 1;
        p.ops.push_back(op);
        return true;
      }
      case RL_H_PLANAR: {
        DevOp op{};
        op.code = OP_PLANAR, op.a = h.index, op.b = NONE;
        op.skip = (uint32_t)p.ops.size() + 1;
        p.ops.push_back(op);
        p.has_planars = true;
        return true;
      }
      case RL_H_LIST: {
        if (list_busy[h.index]) return fail("cycle through a list");
        const rl_list &l = d.lists[h.index];
        if ((uint64_t)l.first + l.count > d.n_list_items) return fail("list range out of bounds");
        list_busy[h.index] = 1;
        for (uint32_t i = 0; i < l.count; i++)
          if (!emit(d.list_items[l.first + i], depth + 1)) return false;
        list_busy[h.index] = 0;
        return true;
      }
      case RL_H_BVH: {
        if (bvh_busy[h.index]) return fail("cycle through a bvh node");
        const rl_bvh_node &n = d.bvh_nodes[h.index];
        if (n.n_children < 1 || n.n_children > 2) return fail("bvh node must have 1 or 2 children");
        for (uint32_t i = 0; i < n.n_children; i++)
          if (!check(n.child[i])) return false;
        uint32_t idx = (uint32_t)p.ops.size();
        DevOp op{};
        std::memcpy(op.box, n.bbox, sizeof op.box);
        uint32_t finite = BOX_FINITE;  // every bound finite and small enough for the filtered (reciprocal) AABB test
        for (int k = 0; k < 6; k++)
          if (!(std::fabs(n.bbox[k]) <= 1e100)) finite = 0;
        bool all_sph = true, all_pl = true;
        for (uint32_t i = 0; i < n.n_children; i++) {
          all_sph &= n.child[i].kind == RL_H_SPHERE;
          all_pl &= n.child[i].kind == RL_H_PLANAR;
        }
        if (all_sph) {
          op.code = OP_BOX_SPH | finite;
          op.a = sphere_payload(n.child[0].index);
          op.b = n.n_children == 2 ? sphere_payload(n.child[1].index) : NONE;
          op.skip = idx + 1;
          p.ops.push_back(op);
          return true;
        }
        if (all_pl) {
          op.code = OP_BOX_PLANAR | finite;
          op.a = n.child[0].index;
          op.b = n.n_children == 2 ? n.child[1].index : NONE;
          op.skip = idx + 1;
          p.ops.push_back(op);
          p.has_planars = true;
          return true;
        }
        op.code = OP_BOX | finite, op.a = op.b = NONE;
        p.ops.push_back(op);
        bvh_busy[h.index] = 1;
        for (uint32_t i = 0; i < n.n_children; i++)
          if (!emit(n.child[i], depth + 1)) return false;
        bvh_busy[h.index] = 0;
        p.ops[idx].skip = (uint32_t)p.ops.size();
        return true;
      }
      case RL_H_MEDIUM: {  // constant_medium.rs: the boundary is evaluated by itself (twice per ray), not folded into the world
        if (in_medium) return fail("a ConstantMedium boundary must not contain another medium");
        const rl_medium &m = d.media[h.index];
        if (m.material >= d.n_materials) return fail("medium material out of range");
        uint32_t begin = (uint32_t)p.ops.size();
        DevOp op{};
        op.code = OP_MEDIUM_BEGIN, op.a = h.index, op.b = NONE;
        p.ops.push_back(op);
        in_medium = true;
        if (!emit(m.boundary, depth + 1)) return false;
        in_medium = false;
        DevOp e{};
        e.code = OP_MEDIUM_END, e.a = h.index, e.b = begin;
        e.skip = (uint32_t)p.ops.size() + 1;
        p.ops.push_back(e);
        p.ops[begin].skip = (uint32_t)p.ops.size();
        p.has_media = true;
        return true;
      }
      case RL_H_TRANSLATE:
      case RL_H_TRANSFORM: {
        bool tr = h.kind == RL_H_TRANSLATE;
        std::vector<uint8_t> &busy = tr ? tr_busy : tf_busy;
        if (busy[h.index]) return fail("cycle through an instance");
        busy[h.index] = 1;
        uint32_t push_pc = (uint32_t)p.ops.size();
        DevOp op{};
        op.code = tr ? OP_PUSH_TRANSLATE : OP_PUSH_TRANSFORM;
        op.a = h.index;
        op.b = push_stack.empty() ? NONE : push_stack.back();  // parent PUSH: the chain back to the world ray
        op.skip = push_pc + 1;
        p.ops.push_back(op);
        p.has_instances = true;
        push_stack.push_back(push_pc);
        if (push_stack.size() > p.max_instance_depth) p.max_instance_depth = (uint32_t)push_stack.size();
        if (push_stack.size() > 8) return fail("instances nested deeper than 8");
        rl_href child = tr ? d.translates[h.index].child : d.transforms[h.index].child;
        if (!emit(child, depth + 1)) return false;
        push_stack.pop_back();
        DevOp po{};
        po.code = tr ? OP_POP_TRANSLATE : OP_POP_TRANSFORM;
        po.a = h.index, po.b = push_pc;  // matching PUSH
        po.skip = (uint32_t)p.ops.size() + 1;
        p.ops.push_back(po);
        busy[h.index] = 0;
        return true;
      }
    }
    return fail("unknown hittable kind");
  }
};

}  // namespace

int compile_rtiow(const rl_rtiow_scene_desc &d, RtiowProgram &p, std::string &err) {
  auto need = [&](const void *ptr, uint32_t n, const char *what) {
    if (n && !ptr) {
      err = std::string("null array: ") + what;
      return false;
    }
    return true;
  };
  if (!need(d.spheres, d.n_spheres, "spheres") || !need(d.planars, d.n_planars, "planars") || !need(d.translates, d.n_translates, "translates") ||
      !need(d.transforms, d.n_transforms, "transforms") || !need(d.bvh_nodes, d.n_bvh_nodes, "bvh_nodes") || !need(d.lists, d.n_lists, "lists") ||
      !need(d.list_items, d.n_list_items, "list_items") || !need(d.materials, d.n_materials, "materials") ||
      !need(d.textures, d.n_textures, "textures") || !need(d.images, d.n_images, "images") || !need(d.perlins, d.n_perlins, "perlins") ||
      !need(d.media, d.n_media, "media"))
    return RL_E_INVALID;
  for (uint32_t i = 0; i < d.n_media; i++) p.media.push_back(d.media[i]);
  if (d.n_spheres > SPH_INDEX) {
    err = "too many spheres";
    return RL_E_INVALID;
  }
  for (uint32_t i = 0; i < d.n_perlins; i++) {  // perlin.rs:17-36: three permutations of 0..255
    const rl_perlin &pn = d.perlins[i];
    for (const uint32_t *perm : {pn.perm_x, pn.perm_y, pn.perm_z})
      for (int k = 0; k < 256; k++)
        if (perm[k] > 255u) {
          err = "perlin permutation entry out of range";
          return RL_E_INVALID;
        }
    p.perlins.push_back(pn);
  }

  // textures / images / materials
  for (uint32_t i = 0; i < d.n_images; i++) {
    const rl_image &im = d.images[i];
    if (!im.rgb || im.width == 0 || im.height == 0) {
      err = "Image has no data";  // texture.rs:64-67 assert
      return RL_E_INVALID;
    }
    DevImage di{im.width, im.height, (uint64_t)p.image_pool.size()};
    p.image_pool.insert(p.image_pool.end(), im.rgb, im.rgb + (size_t)im.width * im.height * 3);
    p.images.push_back(di);
    p.has_images = true;
  }
  for (uint32_t i = 0; i < d.n_textures; i++) {
    const rl_texture &t = d.textures[i];
    DevTexture dt{t.kind, t.even, t.odd, t.image, {t.color[0], t.color[1], t.color[2]}, t.inv_scale};
    if (t.kind == RL_TEX_CHECKER) {
      if (t.even >= d.n_textures || t.odd >= d.n_textures) {
        err = "checker texture child out of range";
        return RL_E_INVALID;
      }
    } else if (t.kind == RL_TEX_IMAGE) {
      if (t.image >= d.n_images) {
        err = "image index out of range";
        return RL_E_INVALID;
      }
    } else if (t.kind == RL_TEX_NOISE) {
      if (t.image >= d.n_perlins) {
        err = "perlin index out of range";
        return RL_E_INVALID;
      }
      p.has_noise = true;
    } else if (t.kind != RL_TEX_SOLID) {
      err = "unknown texture kind";
      return RL_E_INVALID;
    }
    p.textures.push_back(dt);
  }
  // checker nesting must be acyclic and at most MAX_TEX_NEST deep (the device follows it with a bounded loop)
  {
    std::vector<int> nest(d.n_textures, -1);  // -1 unknown, -2 in progress, >=0 nesting depth
    struct Rec {
      static int go(const rl_rtiow_scene_desc &d, std::vector<int> &nest, uint32_t t) {
        if (nest[t] == -2) return -1;  // cycle
        if (nest[t] >= 0) return nest[t];
        if (d.textures[t].kind != RL_TEX_CHECKER) return nest[t] = 0;
        nest[t] = -2;
        int a = go(d, nest, d.textures[t].even), b = go(d, nest, d.textures[t].odd);
        if (a < 0 || b < 0) return -1;
        int r = 1 + (a > b ? a : b);
        if (r > 32) return -1;
        return nest[t] = r;
      }
    };
    for (uint32_t i = 0; i < d.n_textures; i++)
      if (Rec::go(d, nest, i) < 0) {
        err = "checker textures nest cyclically or deeper than 32";
        return RL_E_INVALID;
      }
  }
  for (uint32_t i = 0; i < d.n_materials; i++) {
    const rl_material &m = d.materials[i];
    if (m.kind > RL_MAT_ISOTROPIC) {
      err = "unknown material kind";
      return RL_E_INVALID;
    }
    if ((m.kind == RL_MAT_LAMBERTIAN || m.kind == RL_MAT_DIFFUSE_LIGHT || m.kind == RL_MAT_ISOTROPIC) && m.texture >= d.n_textures) {
      err = "material texture out of range";
      return RL_E_INVALID;
    }
    p.materials.push_back(DevMaterial{m.kind, m.texture, {m.albedo[0], m.albedo[1], m.albedo[2]}, m.fuzz, m.ior});
  }
  // which materials sample an Image (their spheres must carry UVs)
  std::vector<uint8_t> mat_uv(d.n_materials, 0);
  for (uint32_t i = 0; i < d.n_materials; i++) {
    const rl_material &m = d.materials[i];
    if (m.kind != RL_MAT_LAMBERTIAN && m.kind != RL_MAT_DIFFUSE_LIGHT) continue;  // (an Isotropic's hit record carries uv (0, 0))
    std::vector<uint32_t> st{m.texture};
    std::vector<uint8_t> seen(d.n_textures, 0);
    while (!st.empty()) {
      uint32_t t = st.back();
      st.pop_back();
      if (seen[t]) continue;
      seen[t] = 1;
      if (d.textures[t].kind == RL_TEX_IMAGE) mat_uv[i] = 1;
      if (d.textures[t].kind == RL_TEX_CHECKER) st.push_back(d.textures[t].even), st.push_back(d.textures[t].odd);
    }
  }
  for (uint32_t i = 0; i < d.n_spheres; i++) {
    const rl_sphere &s = d.spheres[i];
    if (s.material >= d.n_materials) {
      err = "sphere material out of range";
      return RL_E_INVALID;
    }
    p.sphere_uv.push_back(mat_uv[s.material]);
    if (mat_uv[s.material]) p.has_sphere_uv = true;
    DevSphere ds{};
    for (int k = 0; k < 3; k++) {
      ds.c0[k] = s.center0[k];
      ds.dc[k] = s.moving ? s.center1[k] - s.center0[k] : 0.0;
    }
    ds.r2 = s.radius * s.radius;
    ds.inv_r = 1.0 / s.radius;
    p.spheres.push_back(ds);
    p.sphere_material.push_back(s.material);
  }
  for (uint32_t i = 0; i < d.n_planars; i++) {
    const rl_planar &s = d.planars[i];
    if (s.material >= d.n_materials) {
      err = "planar material out of range";
      return RL_E_INVALID;
    }
    if (s.kind > RL_PLANAR_TRIANGLE) {
      err = "unknown planar kind";
      return RL_E_INVALID;
    }
    DevPlanar dp{};
    std::memcpy(dp.q, s.q, 24), std::memcpy(dp.u, s.u, 24), std::memcpy(dp.v, s.v, 24), std::memcpy(dp.w, s.w, 24);
    std::memcpy(dp.normal, s.normal, 24);
    dp.d = s.d;
    dp.kind = s.kind, dp.material = s.material, dp.has_normals = s.has_normals, dp.has_uvs = s.has_uvs;
    std::memcpy(dp.normals, s.normals, sizeof dp.normals);
    std::memcpy(dp.uvs, s.uvs, sizeof dp.uvs);
    p.planars.push_back(dp);
  }
  p.translates.assign(d.translates, d.translates + d.n_translates);
  p.transforms.assign(d.transforms, d.transforms + d.n_transforms);

  RtiowCompiler c{d, p, err, std::vector<uint8_t>(d.n_bvh_nodes, 0), std::vector<uint8_t>(d.n_lists, 0),
                  std::vector<uint8_t>(d.n_translates, 0), std::vector<uint8_t>(d.n_transforms, 0)};
  if (!c.emit(d.root, 0)) return RL_E_INVALID;
  DevOp end{};
  end.code = OP_END, end.skip = (uint32_t)p.ops.size();
  end.a = end.b = NONE;
  p.ops.push_back(end);
  if (p.ops.size() >= 0x7FFFFFFFu) {
    err = "program too large";
    return RL_E_INVALID;
  }
  return RL_OK;
}

// ---------------------------------------------------------------- RTC
namespace {
struct RtcCompiler {
  const rl_rtc_scene_desc &d;
  RtcProgram &p;
  std::string &err;
  std::vector<uint8_t> g_busy, b_busy, t_busy, c_busy;
  std::vector<uint32_t> enter_stack;
  uint32_t csg_depth = 0;
  bool fail(const std::string &m) {
    err = m;
    return false;
  }
  bool check(rl_oref o) {
    switch (o.kind) {
      case RL_O_TRIANGLE: return o.index < d.n_triangles || fail("triangle index out of range");
      case RL_O_GROUP: return o.index < d.n_groups || fail("group index out of range");
      case RL_O_BOUNDED: return o.index < d.n_boundeds || fail("bounded index out of range");
      case RL_O_TRANSFORMED: return o.index < d.n_transformeds || fail("transformed index out of range");
      case RL_O_SPHERE:
      case RL_O_PLANE:
      case RL_O_CUBE:
      case RL_O_CYLINDER:
      case RL_O_CONE:
        if (o.index >= d.n_shapes) return fail("shape index out of range");
        return d.shapes[o.index].kind == o.kind || fail("shape kind does not match its reference");
      case RL_O_CSG: return o.index < d.n_csgs || fail("csg index out of range");
      default: return fail("unknown object kind");
    }
  }
  void emit_tri(uint32_t idx) {
    if (!p.ops.empty() && p.ops.back().code == ROP_TRIS && p.ops.back().a + p.ops.back().b == idx) {
      p.ops.back().b++;
      return;
    }
    DevOp op{};
    op.code = ROP_TRIS, op.a = idx, op.b = 1, op.skip = 0;
    p.ops.push_back(op);
  }
  bool emit(rl_oref o, uint32_t depth) {
    if (depth > MAX_NEST) return fail("scene graph too deep");
    if (!check(o)) return false;
    switch (o.kind) {
      case RL_O_TRIANGLE:
        emit_tri(o.index);
        return true;
      case RL_O_SPHERE:
      case RL_O_PLANE:
      case RL_O_CUBE:
      case RL_O_CYLINDER:
      case RL_O_CONE: {
        DevOp op{};
        op.code = ROP_SHAPE, op.a = o.index;
        p.ops.push_back(op);
        p.needs_full = true;
        return true;
      }
      case RL_O_CSG: {
        if (c_busy[o.index]) return fail("cycle through a csg");
        const rl_rtc_csg &c = d.csgs[o.index];
        if (c.operation > RL_CSG_DIFFERENCE) return fail("unknown csg operation");
        c_busy[o.index] = 1;
        csg_depth++;
        if (csg_depth > p.max_csg_depth) p.max_csg_depth = csg_depth;
        if (csg_depth > 4) return fail("CSG nesting deeper than 4");
        DevOp op{};
        op.code = ROP_CSG_BEGIN, op.a = o.index;
        p.ops.push_back(op);
        if (!emit(c.left, depth + 1)) return false;
        DevOp mid{};
        mid.code = ROP_CSG_MID, mid.a = o.index;
        p.ops.push_back(mid);
        if (!emit(c.right, depth + 1)) return false;
        DevOp end{};
        end.code = ROP_CSG_END, end.a = o.index;
        p.ops.push_back(end);
        csg_depth--;
        c_busy[o.index] = 0;
        p.needs_full = true;
        return true;
      }
      case RL_O_GROUP: {
        if (g_busy[o.index]) return fail("cycle through a group");
        const rl_rtc_group &g = d.groups[o.index];
        if ((uint64_t)g.first + g.count > d.n_group_items) return fail("group range out of bounds");
        g_busy[o.index] = 1;
        for (uint32_t i = 0; i < g.count; i++)
          if (!emit(d.group_items[g.first + i], depth + 1)) return false;
        g_busy[o.index] = 0;
        return true;
      }
      case RL_O_BOUNDED: {
        if (b_busy[o.index]) return fail("cycle through a bounded");
        const rl_rtc_bounded &b = d.boundeds[o.index];
        uint32_t idx = (uint32_t)p.ops.size();
        DevOp op{};
        op.code = ROP_BOUNDS;
        op.box[0] = b.minimum[0], op.box[1] = b.maximum[0], op.box[2] = b.minimum[1], op.box[3] = b.maximum[1], op.box[4] = b.minimum[2],
        op.box[5] = b.maximum[2];
        p.ops.push_back(op);
        b_busy[o.index] = 1;
        if (!emit(b.child, depth + 1)) return false;
        b_busy[o.index] = 0;
        p.ops[idx].skip = (uint32_t)p.ops.size();
        return true;
      }
      case RL_O_TRANSFORMED: {
        if (t_busy[o.index]) return fail("cycle through a transformed");
        t_busy[o.index] = 1;
        uint32_t enter_pc = (uint32_t)p.ops.size();
        uint32_t parent = enter_stack.empty() ? NONE : enter_stack.back();
        DevOp op{};
        op.code = ROP_ENTER, op.a = o.index, op.b = parent;  // .b: pc of the enclosing ENTER (chain to the world ray)
        p.ops.push_back(op);
        enter_stack.push_back(enter_pc);
        if (enter_stack.size() > p.max_xform_depth) p.max_xform_depth = (uint32_t)enter_stack.size();
        if (enter_stack.size() > 8) return fail("Transformed nesting deeper than 8");
        if (!emit(d.transformeds[o.index].child, depth + 1)) return false;
        enter_stack.pop_back();
        DevOp ex{};
        ex.code = ROP_EXIT, ex.a = o.index, ex.skip = enter_pc, ex.b = parent;
        p.ops.push_back(ex);
        t_busy[o.index] = 0;
        return true;
      }
    }
    return fail("unknown object kind");
  }
};
}  // namespace

int compile_rtc(const rl_rtc_scene_desc &d, RtcProgram &p, std::string &err) {
  auto need = [&](const void *ptr, uint32_t n, const char *what) {
    if (n && !ptr) {
      err = std::string("null array: ") + what;
      return false;
    }
    return true;
  };
  if (!need(d.triangles, d.n_triangles, "triangles") || !need(d.groups, d.n_groups, "groups") || !need(d.group_items, d.n_group_items, "group_items") ||
      !need(d.boundeds, d.n_boundeds, "boundeds") || !need(d.transformeds, d.n_transformeds, "transformeds") ||
      !need(d.materials, d.n_materials, "materials") || !need(d.objects, d.n_objects, "objects") || !need(d.lights, d.n_lights, "lights"))
    return RL_E_INVALID;
  for (uint32_t i = 0; i < d.n_triangles; i++) {
    const rl_rtc_triangle &t = d.triangles[i];
    if (t.material >= d.n_materials) {
      err = "triangle material out of range";
      return RL_E_INVALID;
    }
    DevTri dt{};
    std::memcpy(dt.p1, t.p1, 24), std::memcpy(dt.e1, t.e1, 24), std::memcpy(dt.e2, t.e2, 24);
    std::memcpy(dt.n1, t.n1, 24), std::memcpy(dt.n2, t.n2, 24), std::memcpy(dt.n3, t.n3, 24);
    dt.smooth = t.smooth, dt.material = t.material;
    p.tris.push_back(dt);
  }
  p.xforms.assign(d.transformeds, d.transformeds + d.n_transformeds);
  p.materials.assign(d.materials, d.materials + d.n_materials);
  p.lights.assign(d.lights, d.lights + d.n_lights);
  if (!need(d.shapes, d.n_shapes, "shapes") || !need(d.csgs, d.n_csgs, "csgs") || !need(d.patterns, d.n_patterns, "patterns")) return RL_E_INVALID;
  p.shapes.assign(d.shapes, d.shapes + d.n_shapes);
  p.csgs.assign(d.csgs, d.csgs + d.n_csgs);
  p.patterns.assign(d.patterns, d.patterns + d.n_patterns);
  for (const auto &sh : p.shapes)
    if (sh.material >= d.n_materials) {
      err = "shape material out of range";
      return RL_E_INVALID;
    }
  for (const auto &pt : p.patterns)
    if (pt.kind < RL_PAT_STRIPE || pt.kind > RL_PAT_CHECKER3D) {
      err = "unknown pattern kind";
      return RL_E_INVALID;
    }
  for (const auto &m : p.materials) {
    if (m.reflectivity != 0.0 || m.transparency != 0.0) p.needs_secondary = true;
    if (m.pattern > d.n_patterns) {
      err = "material pattern out of range";
      return RL_E_INVALID;
    }
    if (m.pattern != 0) p.needs_full = true;
  }
  if (p.needs_secondary) p.needs_full = true;
  p.max_reflection_depth = d.max_reflection_depth;
  std::memcpy(p.void_color, d.void_color, 24);
  RtcCompiler c{d, p, err, std::vector<uint8_t>(d.n_groups, 0), std::vector<uint8_t>(d.n_boundeds, 0), std::vector<uint8_t>(d.n_transformeds, 0),
                std::vector<uint8_t>(d.n_csgs, 0)};
  for (uint32_t i = 0; i < d.n_objects; i++)
    if (!c.emit(d.objects[i], 0)) return RL_E_INVALID;
  DevOp end{};
  end.code = ROP_END;
  p.ops.push_back(end);
  return RL_OK;
}

}  // namespace rl
