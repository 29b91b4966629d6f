// RaytracedRenderPath re-hosted on the vhr:: facade (include/vhr_render_graph.hpp).  SURVEY.md section 8 row f4.
// Reference: src/render_paths/raytraced_render_path.{h,cpp}.
#pragma once

#include "vhr_render_graph.hpp"

namespace vhr {

class RaytracedRenderPath : public RenderPath {
public:
    using RenderPath::RenderPath;
    void RegisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;
    void DeregisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;

    // "Alpha test for shadows" (raytraced_render_path.h:15, set through the UI at :80-93 which then calls Rebuild())
    int use_anyhit_shader = 0;

    // the path's composition stage is a raster pass and stays with the integrator: body supplied from outside
    ExternalPassCallback composition_pass;
};

}  // namespace vhr
