// What a C++ integrator of the reference does with this library, end to end and headless:
//   VulkanContext / ResourceManager / RenderGraph / HybridRenderPath  (src/rendering_backend, src/render_graph,
//   src/render_paths/hybrid_render_path.*)  ->  vhr::DeviceContext / ResourceManager / RenderGraph / HybridRenderPath
// The frame loop below is Renderer::Render (renderer.cpp:184-235) minus the window: fill PerFrameData
// (:187-205, zero previous matrices on frame 0, frame_index++), UpdatePerFrameUBO, RenderGraph::Execute.
// The two raster stages that stay with the integrator (G-buffer, composition) are fed by the library's stand-ins.
//
// Build: make -C vulkanhybridrenderer_amd/csrc examples      Run (MI355X): examples/hybrid_frames [frames] [out.ppm] [straight]
// (two frames at 320 x 180, DeviceContext::Resize + RenderPath::Build, then [frames] frames at 640 x 360)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "render_paths.hpp"     // vhr::HybridRenderPath on the facade of include/vhr_render_graph.hpp

namespace {

struct Mat4 { float m[16]; };   // glm column-major: m[col * 4 + row]

Mat4 identity() { Mat4 r{}; r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 r{};
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float s = 0.0f;
            for (int k = 0; k < 4; ++k) s += a.m[k * 4 + row] * b.m[c * 4 + k];
            r.m[c * 4 + row] = s;
        }
    return r;
}
// camera transform = T * Ry(yaw) * Rx(pitch) (scene_loader.cpp:60-69 with roll 0) and its rigid inverse
void camera(const float pos[3], float yaw, float pitch, Mat4 &transform, Mat4 &view) {
    const float cy = std::cos(yaw), sy = std::sin(yaw), cp = std::cos(pitch), sp = std::sin(pitch);
    // R = Ry * Rx, columns
    const float R[3][3] = { { cy, sy * sp, sy * cp }, { 0.0f, cp, -sp }, { -sy, cy * sp, cy * cp } };   // R[row][col]
    transform = identity();
    view = identity();
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) { transform.m[c * 4 + r] = R[r][c]; view.m[c * 4 + r] = R[c][r]; }
    for (int r = 0; r < 3; ++r) transform.m[12 + r] = pos[r];
    for (int r = 0; r < 3; ++r) view.m[12 + r] = -(R[0][r] * pos[0] + R[1][r] * pos[1] + R[2][r] * pos[2]);
}
// VkUtils::InfiniteReverseDepthProjection (vulkan_utils.h:494-503) and its inverse
void projection(float yfov, float aspect, float znear, Mat4 &proj, Mat4 &inv) {
    const float s = 1.0f / std::tan(yfov * 0.5f);
    proj = Mat4{}; inv = Mat4{};
    proj.m[0] = s / aspect; proj.m[5] = s; proj.m[11] = -1.0f; proj.m[14] = znear;
    inv.m[0] = aspect / s; inv.m[5] = 1.0f / s; inv.m[11] = 1.0f / znear; inv.m[14] = -1.0f;
}

void quad(std::vector<vhr::Vertex> &v, std::vector<uint32_t> &idx, const float o[3], const float eu[3], const float ev[3]) {
    const float n[3] = { eu[1] * ev[2] - eu[2] * ev[1], eu[2] * ev[0] - eu[0] * ev[2], eu[0] * ev[1] - eu[1] * ev[0] };
    const float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    const uint32_t base = uint32_t(v.size());
    for (int i = 0; i < 4; ++i) {
        const float a = (i == 1 || i == 2) ? 1.0f : 0.0f, b = (i >= 2) ? 1.0f : 0.0f;
        vhr::Vertex vt{};
        for (int k = 0; k < 3; ++k) { vt.pos[k] = o[k] + a * eu[k] + b * ev[k]; vt.normal[k] = n[k] / len; }
        vt.tangent[0] = 1.0f; vt.tangent[3] = 1.0f;
        vt.uv0[0] = a; vt.uv0[1] = b;
        v.push_back(vt);
    }
    for (uint32_t k : { 0u, 1u, 2u, 0u, 2u, 3u }) idx.push_back(base + k);
}

float half_to_float(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000u) << 16, exp = (h >> 10) & 31u, man = h & 1023u;
    uint32_t bits;
    if (exp == 0) {
        if (!man) bits = sign;
        else { int e = -1; uint32_t m = man; do { ++e; m <<= 1; } while (!(m & 1024u)); bits = sign | uint32_t(127 - 15 - e) << 23 | (m & 1023u) << 13; }
    } else if (exp == 31) bits = sign | 0x7f800000u | man << 13;
    else bits = sign | (exp + 112u) << 23 | man << 13;
    float f; std::memcpy(&f, &bits, 4); return f;
}

}  // namespace

int main(int argc, char **argv) {
    const int frames = argc > 1 ? std::atoi(argv[1]) : 8;
    const char *ppm = argc > 2 ? argv[2] : nullptr;
    const bool checkpoint = !(argc > 3 && std::strcmp(argv[3], "straight") == 0);      // "straight": no save / restore half way
    uint32_t W = 320, H = 180;          // the window opens small; after two frames it is resized to 640 x 360 (below)
    try {
        vhr::DeviceContext context(0, W, H);
        vhr::ResourceManager resource_manager(context);
        vhr::RenderGraph render_graph(context, resource_manager);

        // ---- scene: what SceneLoader hands to ResourceManager::UpdateGeometry (two primitives) ----
        std::vector<vhr::Vertex> vertices;
        std::vector<uint32_t> indices;
        std::vector<vhr::Primitive> primitives;
        auto begin_primitive = [&](float r, float g, float b) {
            vhr::Primitive p{};
            const Mat4 I = identity();
            std::memcpy(p.transform, I.m, sizeof I.m);
            p.material.base_color[0] = r; p.material.base_color[1] = g; p.material.base_color[2] = b; p.material.base_color[3] = 1.0f;
            p.material.base_color_texture = p.material.metallic_roughness_texture = p.material.normal_map = -1;
            p.material.metallic_factor = 0.0f; p.material.roughness_factor = 0.8f;
            p.vertex_offset = uint32_t(vertices.size()); p.index_offset = uint32_t(indices.size());
            return p;
        };
        auto end_primitive = [&](vhr::Primitive p) {
            p.index_count = uint32_t(indices.size()) - p.index_offset;
            for (uint32_t i = p.index_offset; i < indices.size(); ++i) indices[i] -= p.vertex_offset;    // indices are primitive-relative
            primitives.push_back(p);
        };
        {   // floor, facing +y
            vhr::Primitive p = begin_primitive(0.7f, 0.7f, 0.7f);
            const float o[3] = { -10, 0, 10 }, eu[3] = { 20, 0, 0 }, ev[3] = { 0, 0, -20 };
            quad(vertices, indices, o, eu, ev);
            end_primitive(p);
        }
        {   // a 2 m cube standing on the floor
            vhr::Primitive p = begin_primitive(0.8f, 0.3f, 0.2f);
            const float x0 = -1, x1 = 1, y0 = 0, y1 = 2, z0 = -1, z1 = 1;
            const float faces[6][9] = {
                { x0, y0, z1, 2, 0, 0, 0, 2, 0 }, { x1, y0, z0, -2, 0, 0, 0, 2, 0 }, { x1, y0, z1, 0, 0, -2, 0, 2, 0 },
                { x0, y0, z0, 0, 0, 2, 0, 2, 0 }, { x0, y1, z1, 2, 0, 0, 0, 0, -2 }, { x0, y0, z0, 2, 0, 0, 0, 0, 2 } };
            for (auto &f : faces) quad(vertices, indices, f, f + 3, f + 6);
            (void)y1; (void)z1; (void)x1;
            end_primitive(p);
        }
        resource_manager.UpdateGeometry(vertices, indices, primitives);

        // ---- the render path, with the integrator's two raster stages supplied as callbacks ----
        vhr::HybridRenderPath path(context, render_graph, resource_manager);
        path.shadow_mode = vhr::SHADOW_MODE_RAYTRACED;
        path.ambient_occlusion_mode = vhr::AMBIENT_OCCLUSION_MODE_RAYTRACED;
        path.reflection_mode = vhr::REFLECTION_MODE_OFF;
        path.denoise_shadow_and_ao = true;
        uint32_t output = resource_manager.UploadNewStorageImage(W, H, VHR_FORMAT_B8G8R8A8_SRGB);
        path.gbuffer_pass = [&](vhr::DeviceContext &c) {
            vhr::check(c.handle, vhr_standin_gbuffer_with_albedo(c.handle, 0, "Albedo", "World Space Normals and Object IDs",
                                                                 "Motion Vectors and Metallic Roughness", "Depth"), "G-Buffer Pass");
        };
        path.composition_pass = [&](vhr::DeviceContext &c) {
            vhr_composition_desc d{};
            d.shadow_mode = path.shadow_mode; d.ambient_occlusion_mode = path.ambient_occlusion_mode; d.reflection_mode = path.reflection_mode;
            d.albedo_image = "Albedo"; d.normals_image = "World Space Normals and Object IDs";
            d.motion_image = "Motion Vectors and Metallic Roughness"; d.depth_image = "Depth";
            d.shadow_ao_image = "Denoised Raytraced Shadows and Ambient Occlusion";
            d.reflections_image = nullptr;
            d.output_storage_image = int32_t(output);
            vhr::check(c.handle, vhr_standin_composition(c.handle, 0, &d), "Composition Pass");
        };
        path.Build();

        // ---- Renderer::Render, headless ----
        vhr::PerFrameData pfd{};
        Mat4 prev_view{}, prev_proj{};      // zero on frame 0 (function-static zero init, renderer.cpp:188)
        Mat4 proj, proj_inv;
        projection(0.9f, float(W) / float(H), 0.1f, proj, proj_inv);
        uint32_t frame_index = 0;
        auto render_frame = [&](int f) {
            const float pos[3] = { 0.5f + 0.05f * float(f), 3.0f, 8.0f };
            Mat4 transform, view;
            camera(pos, 0.0f, -0.3f, transform, view);
            const Mat4 viewproj_inv = mul(transform, proj_inv);
            std::memcpy(pfd.camera_view, view.m, 64); std::memcpy(pfd.camera_proj, proj.m, 64);
            std::memcpy(pfd.camera_view_inverse, transform.m, 64); std::memcpy(pfd.camera_proj_inverse, proj_inv.m, 64);
            std::memcpy(pfd.camera_viewproj_inverse, viewproj_inv.m, 64);
            std::memcpy(pfd.camera_view_prev_frame, prev_view.m, 64); std::memcpy(pfd.camera_proj_prev_frame, prev_proj.m, 64);
            const float L[3] = { 0.45f, -0.8f, 0.5f };      // from behind the cube towards the camera: the shadow falls in view
            const float ll = std::sqrt(L[0] * L[0] + L[1] * L[1] + L[2] * L[2]);
            for (int k = 0; k < 3; ++k) pfd.directional_light.direction[k] = L[k] / ll;
            for (int k = 0; k < 4; ++k) { pfd.directional_light.color[k] = 1.0f; pfd.directional_light.intensity[k] = 3.0f; }
            const Mat4 I = identity();
            std::memcpy(pfd.directional_light.projview, I.m, 64);
            pfd.display_size[0] = float(W); pfd.display_size[1] = float(H);
            pfd.display_size_inverse[0] = 1.0f / float(W); pfd.display_size_inverse[1] = 1.0f / float(H);
            pfd.frame_index = frame_index++;
            pfd.blue_noise_texture_index = 0;
            resource_manager.UpdatePerFrameUBO(0, pfd);
            render_graph.Execute(0, 0);
            prev_view = view; prev_proj = proj;
        };
        // The reference's one recovery route (renderer.cpp:113-118,146-154): the swapchain is out of date -> `context->Resize();
        // active_render_path->Build();`.  Two frames at the small extent, then the window grows: the context takes the new extent (the graph and the
        // storage pool's images go; geometry and the acceleration structure stay -- UpdateGeometry is NOT called again), the integrator makes its own
        // pool image again, the path registers and builds at the new extent.  Renderer::Render's statics (frame index, last frame's matrices) run on.
        render_frame(-2);
        render_frame(-1);
        W = 640; H = 360;
        context.Resize(W, H);
        output = resource_manager.UploadNewStorageImage(W, H, VHR_FORMAT_B8G8R8A8_SRGB);
        path.Build();
        std::printf("resized to %u x %u after frame %u\n", W, H, frame_index - 1);
        for (int f = 0; f < frames; ++f) {
            render_frame(f);
            if (checkpoint && f == frames / 2) {
                // Checkpoint / resume (the reference has none): the path's cross-frame state out to host memory, the SVGF images thrown away
                // (Rebuild allocates fresh, zeroed ones), the state back in -- the frames that follow are the uninterrupted run's, bit for bit
                // (tests/test_cpp_example.py compares the two runs' checksums).  The blob also carries the caller's half of the state: the last
                // PerFrameData, from which prev_view / prev_proj / frame_index of the next frame follow (renderer.cpp:187-190,202).
                const std::vector<uint8_t> state = path.SaveState();
                path.Rebuild();
                const vhr::PerFrameData last = path.LoadState(state);
                if (std::memcmp(&last, &pfd, sizeof pfd) != 0) throw std::runtime_error("checkpoint: the blob's PerFrameData is not the last frame's");
                std::memcpy(prev_view.m, last.camera_view, 64); std::memcpy(prev_proj.m, last.camera_proj, 64);
                frame_index = last.frame_index + 1;
                std::printf("checkpoint after frame %d: %zu bytes saved and restored\n", f, state.size());
            }
        }
        render_graph.GatherPerformanceStatistics();
        vhr::check(context.handle, vhr_synchronize(context.handle), "vkDeviceWaitIdle");

        // ---- look at the result ----
        std::vector<uint16_t> denoised(size_t(W) * H * 4);
        vhr::check(context.handle, vhr_download_transient_image(context.handle, "Denoised Raytraced Shadows and Ambient Occlusion",
                                                                denoised.data(), denoised.size() * 2), "download");
        std::vector<float> depth(size_t(W) * H);
        vhr::check(context.handle, vhr_download_transient_image(context.handle, "Depth", depth.data(), depth.size() * 4), "download");
        unsigned long long checksum = 1469598103934665603ull;                      // FNV-1a over the denoised image's bits
        for (uint16_t w : denoised) { checksum ^= w; checksum *= 1099511628211ull; }
        std::printf("denoised checksum %016llx\n", checksum);
        double shadow = 0.0, ao = 0.0, covered = 0.0;
        for (size_t i = 0; i < depth.size(); ++i)
            if (depth[i] != 0.0f) { shadow += half_to_float(denoised[4 * i]); ao += half_to_float(denoised[4 * i + 1]); covered += 1.0; }
        shadow /= covered; ao /= covered;
        std::printf("frames %d  covered %.3f  mean shadow %.4f  mean ao %.4f  Raytrace Pass %.3f ms  SVGF Denoise Pass %.3f ms\n", frames,
                    covered / double(depth.size()), shadow, ao, render_graph.PassTimeMs("Raytrace Pass"), render_graph.PassTimeMs("SVGF Denoise Pass"));
        std::vector<uint8_t> bgra(size_t(W) * H * 4);
        vhr::check(context.handle, vhr_download_storage_image(context.handle, int32_t(output), bgra.data(), bgra.size()), "download");
        if (ppm) {
            if (FILE *fp = std::fopen(ppm, "wb")) {
                std::fprintf(fp, "P6\n%u %u\n255\n", W, H);
                for (size_t i = 0; i < size_t(W) * H; ++i) { const uint8_t rgb[3] = { bgra[4 * i + 2], bgra[4 * i + 1], bgra[4 * i] }; std::fwrite(rgb, 1, 3, fp); }
                std::fclose(fp);
            }
        }
        path.DeregisterPath(context, render_graph, resource_manager);
        resource_manager.DestroyStorageImage(output);
        // a cube on a lit floor: part of the floor is in the cube's shadow, most of it is lit; AO darkens the contact edges only
        const bool ok = covered > 0.3 * double(depth.size()) && shadow > 0.5 && shadow < 0.999 && ao > 0.8 && ao <= 1.0;
        std::printf(ok ? "OK\n" : "UNEXPECTED RESULT\n");
        return ok ? 0 : 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "hybrid_frames: %s\n", e.what());
        return 1;
    }
}
