/* Minimal C host for librtw_hip.so (include/rtw_hip.h): the two-sphere scene of the reference's smoke render
 * (/root/reference/test/runtests.jl:194) on every visible MI355X, written as a binary PPM.
 *   gcc -std=c99 -Iinclude examples/render_c.c -Lraytracingweekend.jl_amd/lib -lrtw_hip -Wl,-rpath,$PWD/raytracingweekend.jl_amd/lib -lm -o render_c
 * tests/test_host_abi.py compiles and links it (no GPU needed for that); running it needs a GPU. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "rtw_hip.h"

int main(int argc, char **argv) {
    const int width = argc > 1 ? atoi(argv[1]) : 400, spp = argc > 2 ? atoi(argv[2]) : 64;
    const int height = width * 9 / 16;                                    /* image_width div 16//9 */
    /* scene_2_spheres (src/scenes.jl:2-11): small sphere + ground, both Lambertian */
    const float cx[2] = {0.0f, 0.0f}, cy[2] = {0.0f, -100.5f}, cz[2] = {-1.0f, -1.0f}, r[2] = {0.5f, 100.0f};
    const int32_t kind[2] = {RTW_LAMBERTIAN, RTW_LAMBERTIAN};
    const float ar[2] = {0.7f, 0.8f}, ag[2] = {0.3f, 0.8f}, ab[2] = {0.3f, 0.0f}, param[2] = {0.0f, 0.0f};
    rtw_scene_f32 scene = {2, cx, cy, cz, r, kind, ar, ag, ab, param};
    /* default_camera(SA[0,0,0]) (src/camera.jl:18-36): vfov 90, aspect 16/9, aperture 0, focus 1 */
    rtw_camera_f32 cam = {{0, 0, 0}, {-16.0f / 9.0f, -1.0f, -1.0f}, {32.0f / 9.0f, 0, 0}, {0, 2.0f, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, 0.0f};
    rtw_params p = {0};
    p.width = width; p.height = height; p.spp = spp; p.max_depth = 16; p.seed = 1;
    p.shard_count = 1; p.device = -1; p.gamma = 1;
    p.n_devices = -1;                                                     /* every visible device */
    float *img = (float *)malloc((size_t)width * height * 3 * sizeof(float));
    if (!img) return 2;
    if (rtw_abi_version() != RTW_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 2; }
    int rc = rtw_render_f32(&scene, &cam, &p, img);
    if (rc) { fprintf(stderr, "rtw_render_f32: %d: %s\n", rc, rtw_last_error()); return 1; }
    rtw_stats_t st;
    if (rtw_stats(&st) == 0) fprintf(stderr, "%llu samples, %llu segments, kernel %.3f ms\n", (unsigned long long)st.samples, (unsigned long long)st.segments, st.kernel_ms);
    FILE *f = fopen("render_c.ppm", "wb");
    if (!f) return 2;
    fprintf(f, "P6\n%d %d\n255\n", width, height);
    for (int i = 0; i < height; ++i)
        for (int j = 0; j < width; ++j)
            for (int c = 0; c < 3; ++c) {                                 /* Matrix{RGB{T}} is column-major: pixel (i, j) at (j*H + i)*3 */
                float v = img[((size_t)j * height + i) * 3 + c];
                v = v < 0 ? 0 : (v > 1 ? 1 : v);
                fputc((int)lrintf(v * 255.0f), f);
            }
    fclose(f);
    free(img);
    rtw_shutdown();
    return 0;
}
