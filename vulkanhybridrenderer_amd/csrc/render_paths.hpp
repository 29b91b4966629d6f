// The two render paths re-hosted on the vhr:: facade (include/vhr_render_graph.hpp): what an integrator instantiates.
//   vhr::HybridRenderPath     <- src/render_paths/hybrid_render_path.{h,cpp}      (the hot path: Raytrace Pass + SVGF Denoise Pass)
//   vhr::RaytracedRenderPath  <- src/render_paths/raytraced_render_path.{h,cpp}   
// Settings the reference changes through ImGui radio buttons (then Rebuild()) are plain public members here.
#pragma once

#include <cstdint>
#include <vector>

#include "vhr_render_graph.hpp"

namespace vhr {

// mode values of hybrid_render_path.h:4-20 (also the composition shader's specialization constants)
enum ShadowMode { SHADOW_MODE_RAYTRACED = 0, SHADOW_MODE_RASTERIZED = 1, SHADOW_MODE_OFF = 2 };
enum AmbientOcclusionMode { AMBIENT_OCCLUSION_MODE_RAYTRACED = 0, AMBIENT_OCCLUSION_MODE_SSAO = 1, AMBIENT_OCCLUSION_MODE_OFF = 2 };
enum ReflectionMode { REFLECTION_MODE_RAYTRACED = 0, REFLECTION_MODE_SSR = 1, REFLECTION_MODE_OFF = 2 };

class HybridRenderPath : public RenderPath {
public:
    using RenderPath::RenderPath;
    void RegisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;
    void DeregisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;

    // the raster stages stay with the integrator (G-buffer, composition): bodies supplied from outside
    ExternalPassCallback gbuffer_pass;
    ExternalPassCallback composition_pass;

    // member defaults of hybrid_render_path.h:32-35
    int shadow_mode = SHADOW_MODE_RAYTRACED;
    int ambient_occlusion_mode = AMBIENT_OCCLUSION_MODE_OFF;
    int reflection_mode = REFLECTION_MODE_OFF;
    bool denoise_shadow_and_ao = false;
    int atrous_steps = 5;                              // hybrid_render_path.cpp:299

    // settings of the screen-space alternatives (the ImGui sliders of hybrid_render_path.cpp:422-433); (re)initialised by
    // RegisterPath like the reference does (:139-141, :203-208)
    vhr_ssao_push_constants ssao_push_constants{ 0.75f };
    vhr_ssr_push_constants ssr_push_constants{ 25.0f, 0.1f, 0.5f, 10 };

    // the five persistent SVGF images (pool indices) travel in the push constants, hybrid_render_path.cpp:247-262
    SVGFPushConstants svgf_push_constants{};
    bool svgf_textures_created = false;

    // Checkpoint / resume of the path's cross-frame state (no reference counterpart; vhr_amd.h: vhr_hybrid_save_state).  SaveState returns the blob
    // (header + the last PerFrameData + the five SVGF images as the next frame addresses them); LoadState restores it into this path (same extent) and
    // returns that PerFrameData: the caller continues with view_prev / proj_prev = its view / proj and frame_index + 1 (renderer.cpp:187-190,202).
    std::vector<uint8_t> SaveState();
    PerFrameData LoadState(const std::vector<uint8_t> &blob);
};

class RaytracedRenderPath : public RenderPath {
public:
    using RenderPath::RenderPath;
    void RegisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;
    void DeregisterPath(DeviceContext &context, RenderGraph &render_graph, ResourceManager &resource_manager) override;

    // the path's composition stage is a raster pass and stays with the integrator
    ExternalPassCallback composition_pass;

    // "Alpha test for shadows" (raytraced_render_path.h:15; the UI sets it at raytraced_render_path.cpp:80-93)
    int use_anyhit_shader = 0;
};

}  // namespace vhr
