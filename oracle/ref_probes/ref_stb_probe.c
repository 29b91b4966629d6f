/* TEST INFRASTRUCTURE (oracle/): the reference's vendored stb_image 2.26 (dependencies/stb/stb_image.h) as
 * scene_loader.cpp:277-290 calls it: stbi_load(path, &x, &y, &_, STBI_rgb_alpha).  Writes "<w> <h>\n" and the RGBA8 texels to
 * stdout; committed as tests/golden/ref_stb_decodes.npz next to the encoded files and compared with gltf.py's decoder. */
#define STB_IMAGE_IMPLEMENTATION
#include "stb/stb_image.h"
#include <stdio.h>

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    int x, y, n;
    unsigned char *p = stbi_load(argv[1], &x, &y, &n, STBI_rgb_alpha);
    if (!p) { fprintf(stderr, "stbi_load failed: %s\n", stbi_failure_reason()); return 1; }
    printf("%d %d\n", x, y);
    fwrite(p, 1, (size_t)x * (size_t)y * 4, stdout);
    stbi_image_free(p);
    return 0;
}
