// Camera::render for both crates, on the GPU through the C ABI of include/rl_render.h: the host-side
// call a maintainer of the reference would add beside the CPU render.
//   rtiow::Camera::render / render_from_checkpoint  <- ray-tracing-one-weekend/src/camera.rs:122,136
//   rtc::Camera::render                              <- ray-tracer-challenge/src/scene/camera.rs:93
#include <stdexcept>
#include <string>

#include "rtc_host.hpp"
#include "rtiow_host.hpp"

namespace rtiow {
Canvas Camera::render_internal(uint64_t samples_already_rendered, const Hittable &world) const {
  Flattened f;
  f.root = world.flatten(f);
  rl_rtiow_scene_desc d = f.desc();
  rl_scene *sc = rl_rtiow_scene_create(&d);
  if (!sc) throw std::runtime_error(std::string("rl_rtiow_scene_create: ") + rl_last_error());
  rl_rtiow_camera cam = derived();
  Canvas c{params.samples_per_pixel, params.image_width, image_height, std::vector<double>(params.image_width * image_height * 3)};
  int rc = rl_rtiow_render(sc, &cam, samples_already_rendered, c.data.data(), nullptr);
  rl_scene_destroy(sc);
  if (rc != RL_OK) throw std::runtime_error(std::string("rl_rtiow_render: ") + rl_last_error());
  return c;
}
Canvas Camera::render(const Hittable &world) const { return render_internal(0, world); }  // camera.rs:122
Canvas Camera::render_from_checkpoint(const Hittable &world, const Canvas &checkpoint) const {  // camera.rs:136-143
  return render_internal(checkpoint.samples, world).merge(checkpoint);
}
}  // namespace rtiow

namespace rtc {
Canvas Camera::render(const World &world, const RenderOpts &opts) const {  // scene/camera.rs:93
  Flattened f;
  world.flatten(f);
  rl_rtc_scene_desc d = f.desc();
  rl_scene *sc = rl_rtc_scene_create(&d);
  if (!sc) throw std::runtime_error(std::string("rl_rtc_scene_create: ") + rl_last_error());
  rl_rtc_camera cam = derived();
  Canvas c{hsize, vsize, std::vector<double>(hsize * vsize * 3)};
  int rc = rl_rtc_render(sc, &cam, (uint32_t)opts.anti_aliasing_samples, c.data.data(), nullptr);
  rl_scene_destroy(sc);
  if (rc != RL_OK) throw std::runtime_error(std::string("rl_rtc_render: ") + rl_last_error());
  return c;
}
}  // namespace rtc
