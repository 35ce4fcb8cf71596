// emul_main.cpp -- DEBUG AID: runs render_kernel of rtw_device.hip on the CPU under ASan/UBSan.
// usage: emul_main OBJ W H ns depth   (build: see tests/cpu_emul/Makefile)
#define RTW_HOST_EMUL 1
#include "../../raytracerwin_amd/csrc/rtw_device.hip"
#include "../../raytracerwin_amd/csrc/rtw_host.h"
#include <algorithm>
#include <cstdio>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 6) return 1;
    rtw::HostMesh m;
    std::string err = rtw::load_obj(argv[1], m);
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 2; }
    rtw::build_tree(m);
    rtw::build_quads(m);
    RtwRenderParams p; memset(&p, 0, sizeof p);
    p.width = atoi(argv[2]); p.height = atoi(argv[3]); p.sub_samples = atoi(argv[4]); p.max_bounce = atoi(argv[5]);
    p.world = 1; p.seed = 12345;
    int r0 = argc > 6 ? atoi(argv[6]) : 0, r1 = argc > 7 ? atoi(argv[7]) : p.height;
    p.begin = r0 * p.width; p.count = (r1 - r0) * p.width;
    static RtwSceneDev sc; memset(&sc, 0, sizeof sc);
    sc.n_shapes = 1; sc.prune = argc > 8 ? atoi(argv[8]) : 1;
    std::vector<float> table((size_t)RTW_TABLE_SIZE * 3);
    rtw::fill_unit_table(table.data(), 8);
    std::vector<float> thr(256), lut(256);
    rtw::gamma_thresholds(thr.data()); rtw::texel_lut(lut.data());
    sc.unit_table = table.data(); sc.gamma_thr = thr.data(); sc.texel_lut = lut.data();
    RtwShapeDev& d = sc.shapes[0];
    d.nodes = m.nodes.data(); d.tris = m.tris.data(); d.shade = m.shade.data();
    d.quads = m.quads.data(); d.n_quads = (int)m.quads.size(); d.quad_depth = m.quad_depth;
    sc.traversal = argc > 9 ? atoi(argv[9]) : 1;
    fprintf(stderr, "quads %zu depth %d\n", m.quads.size(), m.quad_depth);
    std::vector<uint32_t> atlas;
    for (size_t t = 0; t < m.textures.size() && t < RTW_DEV_MAX_TEXTURES; t++) {
        if (!m.textures[t].valid) continue;
        d.textures[t].offset = (uint32_t)atlas.size(); d.textures[t].width = m.textures[t].width; d.textures[t].height = m.textures[t].height; d.textures[t].valid = 1;
        atlas.insert(atlas.end(), m.textures[t].rgba8.begin(), m.textures[t].rgba8.end());
    }
    d.texels = atlas.data();
    for (int k = 0; k < 3; k++) { d.bmin[k] = m.bmin[k]; d.bmax[k] = m.bmax[k]; }
    d.n_nodes = (int)m.nodes.size(); d.n_tris = (int)m.tris.size(); d.n_textures = m.n_textures_slots;
    d.has_material = 1; d.n_material_nodes = 1;
    d.material[0].type = 0; d.material[0].r = d.material[0].g = d.material[0].b = 1.0f;
    std::vector<float4> accum((size_t)p.width * p.height); memset(accum.data(), 0, accum.size() * 16);
    std::vector<uint32_t> argb((size_t)p.width * p.height);
    blockDim.x = 256;
    const int grid = (p.count + 255) / 256;
    gridDim.x = (unsigned)grid;
    std::vector<float4> ws((size_t)grid * 256 * (size_t)p.max_bounce * 3);
    const bool hist = getenv("RTW_EMUL_HIST") != nullptr;
    static unsigned long long stats[8]; sc.stats = stats;
    std::vector<unsigned> per_px_box, per_px_tri;
    for (int b = 0; b < grid; b++) {
        blockIdx.x = (unsigned)b;
        for (int t = 0; t < 256; t++) {
            threadIdx.x = (unsigned)t;
            if (hist) {
                const unsigned long long b0 = stats[1], t0 = stats[2];
                render_kernel<true>(&sc, accum.data(), argb.data(), ws.data(), p);
                per_px_box.push_back((unsigned)(stats[1] - b0)); per_px_tri.push_back((unsigned)(stats[2] - t0));
            } else render_kernel<false>(&sc, accum.data(), argb.data(), ws.data(), p);
        }
    }
    if (hist) {
        auto report = [](const char* name, std::vector<unsigned> v) {
            std::vector<unsigned> w; for (unsigned x : v) if (x > 1) w.push_back(x);
            std::sort(w.begin(), w.end());
            unsigned long long sum = 0; for (unsigned x : w) sum += x;
            if (w.empty()) return;
            printf("%s: paths %zu mean %.1f p50 %u p90 %u p99 %u p99.9 %u max %u\n", name, w.size(), (double)sum / w.size(),
                   w[w.size() / 2], w[w.size() * 9 / 10], w[w.size() * 99 / 100], w[(size_t)(w.size() * 0.999)], w.back());
        };
        report("box tests per path", per_px_box);
        report("tri tests per path", per_px_tri);
        // per-wave (64 consecutive pixels) max vs mean, as the one-thread-per-pixel kernel sees it
    }
    double s = 0; for (auto& a : accum) s += a.x + a.y + a.z;
    printf("emul ok: checksum %.6f  tame %llu untame %llu\n", s, g_emul_tame, g_emul_untame);
    return 0;
}
