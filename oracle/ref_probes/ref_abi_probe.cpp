// TEST INFRASTRUCTURE (oracle/): layout probe of the REFERENCE's own shared CPU/GPU header.
// Compiles /root/reference/src/rendering_backend/glsl_common.h (with the reference's vendored glm) where it lies and prints
// sizeof / offsetof / size of every field of every struct it declares as JSON.  The output is committed as
// tests/golden/ref_abi_layout.json (tests/golden/make_ref_pins.py) and tests/test_reference_pins.py compares the product's
// include/vhr_types.h and the numpy mirrors with it field by field.  Built by `make -C oracle ref` into oracle/_ref/ only.
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include "rendering_backend/glsl_common.h"

#define BEGIN(S) std::printf("%s  \"%s\": {\"size\": %zu, \"fields\": {", first_struct ? "" : ",\n", #S, sizeof(S)); first_struct = false; first_field = true;
#define FIELD(S, F) std::printf("%s\"%s\": [%zu, %zu]", first_field ? "" : ", ", #F, offsetof(S, F), sizeof(((S *)0)->F)); first_field = false;
#define END() std::printf("}}");

int main() {
    bool first_struct = true, first_field = true;
    std::printf("{\n");
    BEGIN(DefaultPushConstants) FIELD(DefaultPushConstants, object_id) END()
    BEGIN(HybridPushConstants) FIELD(HybridPushConstants, normal_matrix) FIELD(HybridPushConstants, object_id) END()
    BEGIN(SVGFPushConstants)
        FIELD(SVGFPushConstants, integrated_shadow_and_ao) FIELD(SVGFPushConstants, prev_frame_normals_and_object_ids)
        FIELD(SVGFPushConstants, shadow_and_ao_history) FIELD(SVGFPushConstants, shadow_and_ao_moments_history) FIELD(SVGFPushConstants, atrous_step)
    END()
    BEGIN(SSRPushConstants)
        FIELD(SSRPushConstants, ray_distance) FIELD(SSRPushConstants, step_size) FIELD(SSRPushConstants, thickness) FIELD(SSRPushConstants, bsearch_steps)
    END()
    BEGIN(SSAOPushConstants) FIELD(SSAOPushConstants, radius) END()
    BEGIN(DirectionalLight)
        FIELD(DirectionalLight, projview) FIELD(DirectionalLight, direction) FIELD(DirectionalLight, color) FIELD(DirectionalLight, intensity)
    END()
    BEGIN(PerFrameData)
        FIELD(PerFrameData, camera_view) FIELD(PerFrameData, camera_proj) FIELD(PerFrameData, camera_view_inverse)
        FIELD(PerFrameData, camera_proj_inverse) FIELD(PerFrameData, camera_viewproj_inverse) FIELD(PerFrameData, camera_view_prev_frame)
        FIELD(PerFrameData, camera_proj_prev_frame) FIELD(PerFrameData, directional_light) FIELD(PerFrameData, display_size)
        FIELD(PerFrameData, display_size_inverse) FIELD(PerFrameData, frame_index) FIELD(PerFrameData, blue_noise_texture_index)
    END()
    BEGIN(Vertex) FIELD(Vertex, pos) FIELD(Vertex, normal) FIELD(Vertex, tangent) FIELD(Vertex, uv0) FIELD(Vertex, uv1) END()
    BEGIN(Material)
        FIELD(Material, base_color) FIELD(Material, base_color_texture) FIELD(Material, metallic_roughness_texture) FIELD(Material, normal_map)
        FIELD(Material, metallic_factor) FIELD(Material, roughness_factor) FIELD(Material, alpha_mask) FIELD(Material, alpha_cutoff)
    END()
    BEGIN(Primitive)
        FIELD(Primitive, transform) FIELD(Primitive, material) FIELD(Primitive, vertex_offset) FIELD(Primitive, index_offset) FIELD(Primitive, index_count)
    END()
    // glm's mat4 element order as the reference's C++ side sees it: m[col][row] at float index col * 4 + row
    mat4 m(0.0f);
    m[1][2] = 7.0f;
    const float *f = reinterpret_cast<const float *>(&m);
    int at = -1;
    for (int i = 0; i < 16; ++i) if (f[i] == 7.0f) at = i;
    std::printf(",\n  \"mat4_col1_row2_float_index\": %d\n}\n", at);
    return 0;
}
