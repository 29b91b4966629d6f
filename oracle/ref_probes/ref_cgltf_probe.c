/* TEST INFRASTRUCTURE (oracle/): the reference's vendored cgltf 1.9 (dependencies/cgltf/cgltf.h), called the way
 * SceneLoader::ParseglTF / ParseNode call it (scene_loader.cpp:40-332): cgltf_parse_file + cgltf_load_buffers, then per node
 * cgltf_node_transform_world, per primitive cgltf_accessor_read_float / cgltf_accessor_read_index, the material / texture /
 * sampler / camera / light fields the loader reads.  Prints JSON; committed as tests/golden/ref_gltf_*.json and compared with
 * vulkanhybridrenderer_amd/gltf.py on the same files (tests/test_reference_pins.py). */
#define CGLTF_IMPLEMENTATION
#include "cgltf/cgltf.h"
#include <stdio.h>
#include <string.h>

static void floats(const float *f, int n) { printf("["); for (int i = 0; i < n; ++i) printf("%s%.9g", i ? ", " : "", f[i]); printf("]"); }

static void attribute(const char *name, const cgltf_accessor *acc, int ncomp) {
    printf(", \"%s\": ", name);
    if (!acc) { printf("null"); return; }
    printf("[");
    for (cgltf_size j = 0; j < acc->count; ++j) {
        float v[4] = { 0, 0, 0, 0 };
        cgltf_accessor_read_float(acc, j, v, (cgltf_size)ncomp);
        printf("%s", j ? ", " : ""); floats(v, ncomp);
    }
    printf("]");
}

static void texture(const char *name, const cgltf_texture *t) {
    printf(", \"%s\": ", name);
    if (!t) { printf("null"); return; }
    printf("{\"image_name\": \"%s\", \"image_uri_is_file\": %s, \"sampler\": ", t->image && t->image->name ? t->image->name : "",
           t->image && t->image->uri && strncmp(t->image->uri, "data:", 5) ? "true" : "false");
    if (t->sampler) printf("[%d, %d, %d, %d]}", t->sampler->mag_filter, t->sampler->min_filter, t->sampler->wrap_s, t->sampler->wrap_t);
    else printf("null}");
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    cgltf_options options;
    memset(&options, 0, sizeof options);
    cgltf_data *data = NULL;
    if (cgltf_parse_file(&options, argv[1], &data) != cgltf_result_success) { fprintf(stderr, "parse failed\n"); return 1; }
    if (cgltf_load_buffers(&options, data, argv[1]) != cgltf_result_success) { fprintf(stderr, "load_buffers failed\n"); return 1; }
    printf("{\"nodes\": [");
    for (cgltf_size i = 0; i < data->nodes_count; ++i) {
        cgltf_node *node = &data->nodes[i];
        float world[16];
        cgltf_node_transform_world(node, world);
        printf("%s\n {\"name\": \"%s\", \"world\": ", i ? "," : "", node->name ? node->name : "");
        floats(world, 16);
        if (node->camera && node->camera->type == cgltf_camera_type_perspective)
            printf(", \"camera\": {\"yfov\": %.9g, \"aspect_ratio\": %.9g, \"znear\": %.9g}", node->camera->data.perspective.yfov,
                   node->camera->data.perspective.aspect_ratio, node->camera->data.perspective.znear);
        if (node->light) { printf(", \"light\": {\"directional\": %s, \"color\": ", node->light->type == cgltf_light_type_directional ? "true" : "false"); floats(node->light->color, 3); printf("}"); }
        if (node->mesh) {
            printf(", \"primitives\": [");
            for (cgltf_size k = 0; k < node->mesh->primitives_count; ++k) {
                cgltf_primitive *prim = &node->mesh->primitives[k];
                const cgltf_accessor *pos = NULL, *nrm = NULL, *tan = NULL, *uv0 = NULL, *uv1 = NULL;
                for (cgltf_size j = 0; j < prim->attributes_count; ++j) {
                    const cgltf_attribute *a = &prim->attributes[j];
                    if (a->type == cgltf_attribute_type_position) pos = a->data;
                    else if (a->type == cgltf_attribute_type_normal) nrm = a->data;
                    else if (a->type == cgltf_attribute_type_tangent) tan = a->data;
                    else if (a->type == cgltf_attribute_type_texcoord) { if (a->index == 0) uv0 = a->data; else if (a->index == 1) uv1 = a->data; }
                }
                printf("%s\n  {\"triangles\": %s", k ? "," : "", prim->type == cgltf_primitive_type_triangles ? "true" : "false");
                attribute("pos", pos, 3); attribute("normal", nrm, 3); attribute("tangent", tan, 4); attribute("uv0", uv0, 2); attribute("uv1", uv1, 2);
                printf(", \"indices\": ");
                if (prim->indices) {
                    printf("[");
                    for (cgltf_size j = 0; j < prim->indices->count; ++j) printf("%s%u", j ? ", " : "", (unsigned)cgltf_accessor_read_index(prim->indices, j));
                    printf("]");
                } else printf("null");
                const cgltf_material *m = prim->material;
                if (m) {
                    printf(", \"material\": {\"has_pbr\": %s, \"base_color_factor\": ", m->has_pbr_metallic_roughness ? "true" : "false");
                    floats(m->pbr_metallic_roughness.base_color_factor, 4);
                    printf(", \"metallic_factor\": %.9g, \"roughness_factor\": %.9g, \"alpha_mask\": %s, \"alpha_cutoff\": %.9g", m->pbr_metallic_roughness.metallic_factor,
                           m->pbr_metallic_roughness.roughness_factor, m->alpha_mode == cgltf_alpha_mode_mask ? "true" : "false", m->alpha_cutoff);
                    texture("base_color_texture", m->pbr_metallic_roughness.base_color_texture.texture);
                    texture("metallic_roughness_texture", m->pbr_metallic_roughness.metallic_roughness_texture.texture);
                    texture("normal_texture", m->normal_texture.texture);
                    printf("}");
                }
                printf("}");
            }
            printf("]");
        }
        printf("}");
    }
    int directional = 0;
    for (cgltf_size i = 0; i < data->lights_count; ++i) if (data->lights[i].type == cgltf_light_type_directional) ++directional;
    printf("\n], \"directional_lights\": %d}\n", directional);
    cgltf_free(data);
    return 0;
}
