// default_scene.cpp -- the scene RayTracerProgram::SetupScene builds (Src/RayTracerProgram.cpp:467-552: four spheres, a capsule, the
// checkered ground plane and the unitychan mesh), rendered on the GPU by the reference's render loop (UpdateBitmapPixels) at the
// reference's window size (800 x 800, Src/RayTracerProgram.cpp:31-32).  The shapes are listed as data and added in the reference's order.
//   usage: default_scene UNITYCHAN.obj [PASSES=8] [MAXBOUNCE=10] [W=800] [H=800] [OUT.argb]
// Build: g++ -std=c++11 -Iinclude examples/default_scene.cpp -Lraytracerwin_amd -lrtwin -Wl,-rpath,$PWD/raytracerwin_amd
#include <cstdio>
#include <cstdlib>

#include "RayTracerWin.hpp"

namespace {
typedef std::unique_ptr<ISurfaceMaterial> Mat;
Mat Mirror(const RVec3& c = RVec3(1, 1, 1), float fuzz = 0.0f) { return MakeUnique<SurfaceMaterial_Reflective>(c, fuzz); }
Mat Matte(const RVec3& c) { return MakeUnique<SurfaceMaterial_Diffuse>(c); }
Mat Mix(Mat a, Mat b, float f) { return MakeUnique<SurfaceMaterial_Blend>(std::move(a), std::move(b), f); }

struct Ball { RVec3 c; float r; };
const Ball kBalls[4] = { { RVec3(1.5f, 2.5f, -2.0f), 0.9f }, { RVec3(-1.5f, -0.5f, -3.0f), 0.5f }, { RVec3(0.8f, -1.5f, -1.0f), 0.5f },
                         { RVec3(2.8f, -1.2f, -4.0f), 1.5f } };

Mat BallMaterial(int i)
{
    const RVec3 gold(0.95f, 0.75f, 0.1f);
    switch (i) {
    case 0: return Mix(Mirror(), Matte(RVec3(1.0f, 0.5f, 0.1f)), 0.5f);
    case 1: return Matte(RVec3(0.1f, 1.0f, 0.2f));
    case 2: return Mix(Mirror(), Matte(RVec3(0.5f, 0.0f, 0.2f)), 0.5f);
    default: return MakeUnique<SurfaceMaterial_Combine>(Mix(Mirror(gold), Matte(gold), 0.5f), MakeUnique<SurfaceMaterial_Emissive>(gold * 0.5f));
    }
}
}  // namespace

int main(int argc, char** argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s UNITYCHAN.obj [PASSES] [MAXBOUNCE] [W] [H] [OUT.argb]\n", argv[0]); return 2; }
    const int TotalSamplesNum = argc > 2 ? std::atoi(argv[2]) : 8, MaxBounceTimes = argc > 3 ? std::atoi(argv[3]) : 10;
    const int W = argc > 4 ? std::atoi(argv[4]) : 800, H = argc > 5 ? std::atoi(argv[5]) : 800;
    try {
        RtwDevice Device(0);
        RayTracerScene Scene(Device);
        for (int i = 0; i < 4; i++) Scene.AddShape(RSphere::Create(kBalls[i].c, kBalls[i].r), BallMaterial(i));
        Scene.AddShape(RCapsule::Create(RVec3(-1.5f, -1.5f, -1.5f), RVec3(-2.0f, -1.5f, 0.0f), 0.5f),
                       Mix(Mirror(RVec3(0.8f, 0.75f, 0.6f), 0.2f), Matte(RVec3(0.25f, 0.75f, 0.6f)), 0.2f));
        Scene.AddShape(RPlane::Create(RVec3(0.0f, 1.0f, 0.0f), RVec3(0.0f, -2.0f, 0.0f)),           // the ground
                       Mix(Mirror(RVec3(1, 1, 1), 0.1f), MakeUnique<SurfaceMaterial_DiffuseChecker>(), 0.5f));
        Scene.AddShape(RMeshShape::Create(argv[1]), Mix(Mirror(RVec3(1, 1, 1), 0.2f), Matte(RVec3(1.0f, 1.0f, 1.0f)), 1.0f));
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
