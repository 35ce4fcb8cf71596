// progressive.cpp -- the reference's render thread (UpdateBitmapPixels, Src/RayTracerProgram.cpp:270-422) on the GPU:
// preview pass, N accumulated passes with the reference's progress line, Output_<spp>spp_<date>.png in SavedImages/.
//   usage: progressive MESH.obj WIDTH HEIGHT PASSES MAXBOUNCE [OUT.argb]
// Build: g++ -std=c++11 -Iinclude examples/progressive.cpp -Lraytracerwin_amd -lrtwin -Wl,-rpath,$PWD/raytracerwin_amd
#include <cstdio>
#include <cstdlib>

#include "RayTracerWin.hpp"

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s MESH.obj W H PASSES MAXBOUNCE [OUT.argb]\n", argv[0]); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), TotalSamplesNum = std::atoi(argv[4]), MaxBounceTimes = std::atoi(argv[5]);
    try {
        RtwDevice Device(0);
        RayTracerScene Scene(Device);
        Scene.AddShape(RMeshShape::Create(argv[1]), MakeUnique<SurfaceMaterial_Diffuse>(RVec3(1.0f, 1.0f, 1.0f)));
        ColorBuffer Buffer(Device, W, H);
        const std::string Saved = UpdateBitmapPixels(Device, Scene, Buffer, TotalSamplesNum, MaxBounceTimes);
        if (argc > 6) {
            const std::vector<Pixel> bitcolor = Buffer.bitcolor();
            FILE* f = std::fopen(argv[6], "wb"); std::fwrite(bitcolor.data(), 4, bitcolor.size(), f); std::fclose(f);
        }
        std::printf("saved: %s\n", Saved.c_str());
    } catch (const RtwFailure& e) {
        std::fprintf(stderr, "%s (code %d)\n", e.what(), e.code);
        return 1;
    }
    return 0;
}
