// HybridRenderPath re-hosted on the vhr:: facade (include/vhr_render_graph.hpp).
// Reference: src/render_paths/hybrid_render_path.{h,cpp}.
#pragma once

#include "vhr_render_graph.hpp"

namespace vhr {

// hybrid_render_path.h:4-20
enum ShadowMode { SHADOW_MODE_RAYTRACED = 0, SHADOW_MODE_RASTERIZED = 1, SHADOW_MODE_OFF = 2 };
enum AmbientOcclusionMode { AMBIENT_OCCLUSION_MODE_RAYTRACED = 0, AMBIENT_OCCLUSION_MODE_SSAO = 1, AMBIENT_OCCLUSION_MODE_OFF = 2 };
enum ReflectionMode { REFLECTION_MODE_RAYTRACED = 0, REFLECTION_MODE_SSR = 1, REFLECTION_MODE_OFF = 2 };

class HybridRenderPath : public RenderPath {
public:
    using RenderPath::RenderPath;
    void RegisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;
    void DeregisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;

    // member defaults of hybrid_render_path.h:32-35 (set through the UI in the reference, :394-441)
    int shadow_mode = SHADOW_MODE_RAYTRACED;
    int ambient_occlusion_mode = AMBIENT_OCCLUSION_MODE_OFF;
    int reflection_mode = REFLECTION_MODE_OFF;
    bool denoise_shadow_and_ao = false;
    int atrous_steps = 5;                              // hybrid_render_path.cpp:299

    // the raster stages stay with the integrator (G-buffer, composition): bodies supplied from outside
    ExternalPassCallback gbuffer_pass;
    ExternalPassCallback composition_pass;

    SVGFPushConstants svgf_push_constants{};
    bool svgf_textures_created = false;
};

}  // namespace vhr
