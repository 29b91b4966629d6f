// TEST INFRASTRUCTURE (oracle/): the reference's vendored glm 0.9.9.8 (dependencies/glm) under the reference's own
// configuration (pch.h:37-38 GLM_FORCE_DEPTH_ZERO_TO_ONE, GLM_FORCE_RADIANS), called the way the reference calls it:
//   scene_loader.cpp:58-66  camera:  extractEulerAngleYXZ, yawPitchRoll, translate, inverse
//   scene_loader.cpp:74-94  light:   decompose -> quat, rotate(quat, (0,0,-1)), ortho(-8,8,-8,8,12,0.1) * lookAt(-d*12, 0, +y)
//   renderer.cpp:191-196    frame:   inverse(proj), inverse(proj * view)
// Reads one case per line from stdin, prints one line of results per case (hex floats would be exact; %.9g round-trips fp32).
// Outputs are committed as tests/golden/ref_glm_cases.json and compared with vulkanhybridrenderer_amd/camera.py / gltf.py.
#define GLM_FORCE_DEPTH_ZERO_TO_ONE
#define GLM_FORCE_RADIANS
#define GLM_ENABLE_EXPERIMENTAL
#include <cstdio>
#include <cstring>
#include "glm/vec3.hpp"
#include "glm/vec4.hpp"
#include "glm/mat4x4.hpp"
#include "glm/ext/matrix_clip_space.hpp"
#include "glm/gtx/matrix_decompose.hpp"
#include "glm/ext/matrix_transform.hpp"
#include "glm/gtc/matrix_access.hpp"
#include "glm/gtc/matrix_inverse.hpp"
#include "glm/gtc/type_ptr.hpp"
#include "glm/gtx/euler_angles.hpp"
#include "glm/gtx/quaternion.hpp"

static bool read_floats(const char *&p, float *out, int n) {
    for (int i = 0; i < n; ++i) { int used = 0; if (std::sscanf(p, "%f%n", &out[i], &used) != 1) return false; p += used; }
    return true;
}
static void print_mat(const glm::mat4 &m) { const float *f = glm::value_ptr(m); for (int i = 0; i < 16; ++i) std::printf(" %.9g", f[i]); }

int main() {
    char line[4096];
    while (std::fgets(line, sizeof line, stdin)) {
        char op[32]; int used = 0;
        if (std::sscanf(line, "%31s%n", op, &used) != 1) continue;
        const char *p = line + used;
        float a[32];
        if (!std::strcmp(op, "ortho_lookat") && read_floats(p, a, 3)) {            // scene_loader.cpp:85-94
            glm::mat4 light_perspective = glm::ortho(-8.0f, 8.0f, -8.0f, 8.0f, 12.0f, 0.1f);
            glm::vec3 light_direction(a[0], a[1], a[2]);
            glm::mat4 light_view = glm::lookAt(-light_direction * 12.0f, glm::vec3(0.0f, 0.0f, 0.0f), glm::vec3(0.0f, 1.0f, 0.0f));
            std::printf("ortho_lookat"); print_mat(light_perspective * light_view); std::printf("\n");
        } else if (!std::strcmp(op, "camera") && read_floats(p, a, 16)) {          // scene_loader.cpp:58-66
            glm::mat4 transform = glm::make_mat4(a);
            float yaw, pitch, roll;
            glm::extractEulerAngleYXZ(transform, yaw, pitch, roll);
            glm::mat4 R = glm::yawPitchRoll(yaw, pitch, roll);
            glm::mat4 T = glm::translate(glm::mat4(1.0f), glm::vec3(glm::column(transform, 3)));
            transform = T * R;
            std::printf("camera %.9g %.9g %.9g", yaw, pitch, roll); print_mat(transform); print_mat(glm::inverse(transform)); std::printf("\n");
        } else if (!std::strcmp(op, "lightdir") && read_floats(p, a, 16)) {        // scene_loader.cpp:74-86
            glm::mat4 node_transform = glm::make_mat4(a);
            glm::vec3 translation, skew, scale; glm::quat rot; glm::vec4 persp;
            glm::decompose(node_transform, scale, rot, translation, skew, persp);
            glm::vec3 d = glm::normalize(glm::rotate(rot, glm::vec3(0.0f, 0.0f, -1.0f)));
            std::printf("lightdir %.9g %.9g %.9g\n", d.x, d.y, d.z);
        } else if (!std::strcmp(op, "inverse") && read_floats(p, a, 16)) {         // renderer.cpp:194
            std::printf("inverse"); print_mat(glm::inverse(glm::make_mat4(a))); std::printf("\n");
        } else if (!std::strcmp(op, "inverse_product") && read_floats(p, a, 32)) { // renderer.cpp:195 inverse(proj * view)
            std::printf("inverse_product"); print_mat(glm::inverse(glm::make_mat4(a) * glm::make_mat4(a + 16))); std::printf("\n");
        } else if (!std::strcmp(op, "yaw_pitch_roll") && read_floats(p, a, 3)) {
            std::printf("yaw_pitch_roll"); print_mat(glm::yawPitchRoll(a[0], a[1], a[2])); std::printf("\n");
        } else {
            std::printf("error %s\n", op);
        }
    }
    return 0;
}
