// render_mesh.cpp -- SetupScene-style host code written against the reference's names (include/RayTracerWin.hpp),
// running the hot path on the GPU through librtwin.so.
//   usage: render_mesh MESH.obj WIDTH HEIGHT PASSES MAXBOUNCE OUT.png [OUT.argb]
// Build: g++ -std=c++11 -Iinclude examples/render_mesh.cpp -Lraytracerwin_amd -lrtwin -Wl,-rpath,$PWD/raytracerwin_amd
#include <cstdio>
#include <cstdlib>

#include "RayTracerWin.hpp"

int main(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: %s MESH.obj W H PASSES MAXBOUNCE OUT.png [OUT.argb]\n", argv[0]); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), Passes = std::atoi(argv[4]), MaxBounceTimes = std::atoi(argv[5]);
    try {
        RtwDevice Device(0);
        RayTracerScene Scene(Device);
        // the mesh line of RayTracerProgram::SetupScene (Src/RayTracerProgram.cpp:546-551)
        Scene.AddShape(RMeshShape::Create(argv[1]),
            MakeUnique<SurfaceMaterial_Blend>(
                MakeUnique<SurfaceMaterial_Reflective>(RVec3(1, 1, 1), 0.2f),
                MakeUnique<SurfaceMaterial_Diffuse>(RVec3(1.0f, 1.0f, 1.0f)),
                1.0f));
        ColorBuffer Buffer(Device, W, H);
        const int MaxBufferIdx = W * H - 1, NumTaskRows = 10;
        for (int Sample = 0; Sample < Passes; Sample++) {
            for (int i = 0; i < H; i += NumTaskRows) {              // the task split of UpdateBitmapPixels (Src/RayTracerProgram.cpp:320-327)
                const int Start = i * W;
                const int End = (i + NumTaskRows) * W - 1 < MaxBufferIdx ? (i + NumTaskRows) * W - 1 : MaxBufferIdx;
                ThreadWorker_Render(Scene, Buffer, Start, End, MaxBounceTimes, RenderOption(), Sample);
            }
        }
        Device.Synchronize();
        const std::vector<Pixel> bitcolor = Buffer.bitcolor();
        if (!RTexture::SaveBufferToPNG(argv[6], bitcolor.data(), W, H)) { std::fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
        if (argc > 7) { FILE* f = std::fopen(argv[7], "wb"); std::fwrite(bitcolor.data(), 4, bitcolor.size(), f); std::fclose(f); }
        std::printf("rendered %dx%d, %d passes -> %s\n", W, H, Passes, argv[6]);
    } catch (const RtwFailure& e) {
        std::fprintf(stderr, "%s (code %d)\n", e.what(), e.code);
        return 1;
    }
    return 0;
}
