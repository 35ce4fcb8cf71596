// progressive.cpp -- the reference's render thread (UpdateBitmapPixels, Src/RayTracerProgram.cpp:270-422) on the GPU: preview pass, N accumulated
// passes with the reference's progress line (also the window title), Output_<spp>spp_<date>.png in SavedImages/, and the frames presented
// to a RenderWindow-shaped sink (Src/Linux/RenderWindow_X11.h:11-30) that counts them here.
//   usage: progressive MESH.obj WIDTH HEIGHT PASSES MAXBOUNCE [OUT.argb [PASSES_PER_UPDATE [RANK WORLD IDFILE]]]
// Several ranks (one process per GPU): rank 0 makes the RCCL id and leaves it in IDFILE, the others read it; every rank renders its 10-row tasks,
// rank 0 gathers before every present and before it saves the image.
// Build: g++ -std=c++11 -Iinclude examples/progressive.cpp -Lraytracerwin_amd -lrtwin -Wl,-rpath,$PWD/raytracerwin_amd
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "RayTracerWin.hpp"

#include <algorithm>
#include <vector>
struct Seen { int frames; int titles; unsigned long long checksum; std::vector<double> title_ms; };
static void OnFrame(void* user, const Pixel* px, int w, int h)
{
    Seen* s = (Seen*)user;
    s->frames++;
    unsigned long long sum = 0;
    for (int i = 0; i < w * h; i++) sum += px[i] & 0xFFFFFFu;
    s->checksum = sum;
}
static void OnTitle(void* user, const char*)
{
    Seen* s = (Seen*)user;
    s->titles++;
    s->title_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
// RTW_EXAMPLE_DEVICE_SINK=1: the window takes the image where it lies, in device memory (an interop surface / encoder would read it there): no host copy per present
static void OnDeviceFrame(void* user, const void* device_pixels, int, int) { if (device_pixels) ((Seen*)user)->frames++; }

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: %s MESH.obj W H PASSES MAXBOUNCE [OUT.argb [PASSES_PER_UPDATE [RANK WORLD IDFILE]]]\n", argv[0]); return 2; }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), TotalSamplesNum = std::atoi(argv[4]), MaxBounceTimes = std::atoi(argv[5]);
    try {
        RtwDevice Device(0);
        RayTracerScene Scene(Device);
        Scene.AddShape(RMeshShape::Create(argv[1]), MakeUnique<SurfaceMaterial_Diffuse>(RVec3(1.0f, 1.0f, 1.0f)));
        ColorBuffer Buffer(Device, W, H);
        // the display hook, as RayTracerProgram::Run sets its window up (Src/RayTracerProgram.cpp:445-455)
        std::vector<Pixel> Shown((size_t)W * H);
        Seen seen; seen.frames = 0; seen.titles = 0; seen.checksum = 0ull;
        RenderWindow Window;
        Window.Create(W, H);
        Window.SetRenderBufferParameters(W, H, Shown.data());
        Window.SetSinks(OnFrame, OnTitle, &seen);
        const bool DeviceSink = std::getenv("RTW_EXAMPLE_DEVICE_SINK") != nullptr;
        if (DeviceSink) Window.SetDeviceSink(OnDeviceFrame, &seen);
        RtwProgressive Run;
        Run.TotalSamplesNum = TotalSamplesNum; Run.MaxBounceTimes = MaxBounceTimes; Run.Window = &Window;
        Run.PassesPerUpdate = argc > 7 ? std::atoi(argv[7]) : 1;
        rtw_comm* Comm = nullptr;
        if (argc > 10 && std::atoi(argv[9]) > 1) {      // several ranks: the id travels through a file (any channel will do: MPI, a socket ...)
            Run.Rank = std::atoi(argv[8]); Run.World = std::atoi(argv[9]);
            uint8_t Id[RTW_COMM_ID_BYTES];
            const std::string IdFile = argv[10];
            if (Run.Rank == 0) {
                RtwCheck(rtw_comm_unique_id(Id));
                FILE* f = std::fopen((IdFile + ".tmp").c_str(), "wb"); std::fwrite(Id, 1, sizeof Id, f); std::fclose(f);
                std::rename((IdFile + ".tmp").c_str(), IdFile.c_str());
            } else {
                FILE* f = nullptr;
                for (int tries = 0; tries < 6000 && !(f = std::fopen(IdFile.c_str(), "rb")); tries++) std::this_thread::sleep_for(std::chrono::milliseconds(10));
                if (!f || std::fread(Id, 1, sizeof Id, f) != sizeof Id) { std::fprintf(stderr, "rank %d: no communicator id in %s\n", Run.Rank, IdFile.c_str()); return 3; }
                std::fclose(f);
            }
            RtwCheck(rtw_comm_create(Device.Get(), Id, Run.Rank, Run.World, &Comm));
            Run.Comm = Comm;
            if (Run.Rank != 0) { Run.Window = nullptr; Run.Quiet = true; }
        }
        if (std::getenv("RTW_EXAMPLE_QUIET")) Run.Quiet = true;
        const std::chrono::steady_clock::time_point T0 = std::chrono::steady_clock::now();
        const std::string Saved = UpdateBitmapPixels(Device, Scene, Buffer, Run);
        const double TotalMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T0).count();
        if (Run.Rank == 0 && seen.title_ms.size() > 2) {     // the rhythm of the updates as the window saw them: time from one title to the next
            std::vector<double> gaps;
            for (size_t i = 1; i < seen.title_ms.size(); i++) gaps.push_back(seen.title_ms[i] - seen.title_ms[i - 1]);
            std::sort(gaps.begin(), gaps.end());
            std::printf("loop: %.3f ms for %d passes, %d per update; %.4f ms per update end to end, median over %d updates (render + synchronise + title + present%s)\n", TotalMs, TotalSamplesNum,
                        Run.PassesPerUpdate, gaps[gaps.size() / 2], (int)gaps.size(), DeviceSink ? " of the device image" : " through the host buffer");
        }
        if (Comm) RtwCheck(rtw_comm_destroy(Comm));
        if (Run.Rank != 0) return 0;
        Window.RunWindowLoop();
        if (argc > 6) {
            const std::vector<Pixel> bitcolor = Buffer.bitcolor();
            FILE* f = std::fopen(argv[6], "wb"); std::fwrite(bitcolor.data(), 4, bitcolor.size(), f); std::fclose(f);
            unsigned long long sum = 0;
            for (size_t i = 0; i < bitcolor.size(); i++) sum += bitcolor[i] & 0xFFFFFFu;
            std::printf("window: %d frames presented, %d titles, last frame %s the final image; title: %s\n", seen.frames, seen.titles,
                        DeviceSink ? "stayed on the device, not compared with" : (sum == seen.checksum ? "equals" : "DIFFERS FROM"), Window.GetTitle().c_str());
        }
        std::printf("saved: %s\n", Saved.c_str());
    } catch (const RtwFailure& e) {
        std::fprintf(stderr, "%s (code %d)\n", e.what(), e.code);
        return 1;
    }
    return 0;
}
