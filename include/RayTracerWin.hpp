// RayTracerWin.hpp -- header-only C++ facade over the C ABI (include/rtwin.h) that keeps the reference's
// names for the hot path, so that SetupScene-style code and the image-write call of
// aosyang/RayTracerWin compile against librtwin.so with the same spelling:
//
//     RayTracerScene::AddShape(RMeshShape::Create(path), MakeUnique<SurfaceMaterial_*>(...))
//                                                     Src/RayTracerScene.cpp:25, Src/MeshShape.h:23
//     RayTracerScene::FindIntersectionWithScene       Src/RayTracerScene.cpp:99
//     ThreadWorker_Render(begin, end, MaxBounceCount, RenderOption)   Src/RayTracerProgram.cpp:131
//     RTexture::SaveBufferToPNG                       Src/Texture.cpp:201
//     MakePixelColor / LinearToGamma / MakeUint32Color / GetUint32Color*   Src/ColorBuffer.h:34-109
//
// What differs from the reference, on purpose: the scene, the frame size and the random seed are explicit
// (the reference reaches them through globals and compile-time constants), errors are reported (exceptions
// carrying rtw_last_error()) instead of being logged and ignored, and nothing here runs on the CPU.
#pragma once

#include <chrono>
#include <cmath>
#include <cstdio>
#include <ctime>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rtwin.h"

typedef unsigned int UINT32;
typedef UINT32 Pixel;

class RVec3 {
public:
    float x, y, z;
    RVec3() : x(0.0f), y(0.0f), z(0.0f) {}
    RVec3(float _x, float _y, float _z) : x(_x), y(_y), z(_z) {}
    RVec3 operator*(float s) const { return RVec3(x * s, y * s, z * s); }      // Src/RVector.h
};

struct RtwFailure : std::runtime_error {
    int code;
    RtwFailure(int c) : std::runtime_error(std::string("librtwin: ") + rtw_last_error()), code(c) {}
};
inline void RtwCheck(int rc) { if (rc < 0) throw RtwFailure(rc); }

// ---- Src/ColorBuffer.h ---------------------------------------------------------------------------------------
inline UINT32 MakeUint32Color(unsigned char r, unsigned char g, unsigned char b, unsigned char a) { return (UINT32)(a << 24 | r << 16 | g << 8 | b); }
inline unsigned char GetUint32ColorRed(UINT32 c) { return (unsigned char)((c >> 16) & 0xFF); }
inline unsigned char GetUint32ColorGreen(UINT32 c) { return (unsigned char)((c >> 8) & 0xFF); }
inline unsigned char GetUint32ColorBlue(UINT32 c) { return (unsigned char)(c & 0xFF); }
inline RVec3 LinearToGamma(const RVec3& c) { const float e = 1.0f / 2.2f; return RVec3(powf(c.x, e), powf(c.y, e), powf(c.z, e)); }
inline RVec3 GammaToLinear(const RVec3& c) { const float e = 2.2f; return RVec3(powf(c.x, e), powf(c.y, e), powf(c.z, e)); }
inline UINT32 MakePixelColor(const RVec3& c)
{
    auto q = [](float v) { v = v > 0.0f ? v : 0.0f; v = v < 1.0f ? v : 1.0f; return (int)(v * 255); };
    return MakeUint32Color((unsigned char)q(c.x), (unsigned char)q(c.y), (unsigned char)q(c.z), 255);
}

// ---- Src/SurfaceMaterials.h: the same constructors, flattened into rtw_material_node[] -------------------------------
class ISurfaceMaterial {
public:
    virtual ~ISurfaceMaterial() {}
    virtual int Flatten(std::vector<rtw_material_node>& out) const = 0;
protected:
    static rtw_material_node Node(int type, const RVec3& c, float param, int a = 0, int b = 0)
    {
        rtw_material_node n; n.type = type; n.r = c.x; n.g = c.y; n.b = c.z; n.param = param; n.child_a = a; n.child_b = b; n.pad = 0;
        return n;
    }
};
class RtwLeafMaterial : public ISurfaceMaterial {
public:
    int Flatten(std::vector<rtw_material_node>& out) const override { out.push_back(Node(Type, Color, Param)); return (int)out.size() - 1; }
protected:
    RtwLeafMaterial(int type, const RVec3& c, float param) : Type(type), Color(c), Param(param) {}
private:
    int Type; RVec3 Color; float Param;
};
class SurfaceMaterial_Diffuse : public RtwLeafMaterial {
public:
    SurfaceMaterial_Diffuse(const RVec3 InAlbedo = RVec3(1.0f, 1.0f, 1.0f)) : RtwLeafMaterial(RTW_MAT_DIFFUSE, InAlbedo, 0.0f) {}
};
class SurfaceMaterial_DiffuseChecker : public RtwLeafMaterial {
public:
    SurfaceMaterial_DiffuseChecker(const RVec3 InAlbedo = RVec3(1.0f, 1.0f, 1.0f), float InPatternSize = 5.0f) : RtwLeafMaterial(RTW_MAT_DIFFUSE_CHECKER, InAlbedo, InPatternSize) {}
};
class SurfaceMaterial_Reflective : public RtwLeafMaterial {
public:
    SurfaceMaterial_Reflective(const RVec3 InAlbedo = RVec3(1.0f, 1.0f, 1.0f), float InFuzziness = 0.0f) : RtwLeafMaterial(RTW_MAT_REFLECTIVE, InAlbedo, InFuzziness) {}
};
class SurfaceMaterial_Emissive : public RtwLeafMaterial {
public:
    SurfaceMaterial_Emissive(const RVec3 InColor) : RtwLeafMaterial(RTW_MAT_EMISSIVE, InColor, 0.0f) {}
};
class SurfaceMaterial_Null : public RtwLeafMaterial {
public:
    SurfaceMaterial_Null() : RtwLeafMaterial(RTW_MAT_NULL, RVec3(0.0f, 0.0f, 0.0f), 0.0f) {}
};

class SurfaceMaterial_Blend : public ISurfaceMaterial {
public:
    SurfaceMaterial_Blend(std::unique_ptr<ISurfaceMaterial> InMaterialA, std::unique_ptr<ISurfaceMaterial> InMaterialB, float InBlendFactor)
        : A(std::move(InMaterialA)), B(std::move(InMaterialB)), Factor(InBlendFactor) {}
    int Flatten(std::vector<rtw_material_node>& out) const override
    {
        const int me = (int)out.size();
        out.push_back(Node(RTW_MAT_BLEND, RVec3(), Factor));
        const int ia = A->Flatten(out), ib = B->Flatten(out);       // may reallocate `out`: index afterwards
        out[(size_t)me].child_a = ia; out[(size_t)me].child_b = ib;
        return me;
    }
private:
    std::unique_ptr<ISurfaceMaterial> A, B; float Factor;
};
class SurfaceMaterial_Combine : public ISurfaceMaterial {
public:
    SurfaceMaterial_Combine(std::unique_ptr<ISurfaceMaterial> InMaterialA, std::unique_ptr<ISurfaceMaterial> InMaterialB)
        : A(std::move(InMaterialA)), B(std::move(InMaterialB)) {}
    int Flatten(std::vector<rtw_material_node>& out) const override
    {
        const int me = (int)out.size();
        out.push_back(Node(RTW_MAT_COMBINE, RVec3(), 0.0f));
        const int ia = A->Flatten(out), ib = B->Flatten(out);
        out[(size_t)me].child_a = ia; out[(size_t)me].child_b = ib;
        return me;
    }
private:
    std::unique_ptr<ISurfaceMaterial> A, B;
};

template <typename T, typename... Args>
std::unique_ptr<T> MakeUnique(Args&&... args) { return std::unique_ptr<T>(new T(std::forward<Args>(args)...)); }

// ---- Src/RRay.h ---------------------------------------------------------------------------------------------------------
struct RayHitResult {
    RVec3 HitPosition, HitNormal; float Distance; RVec3 SampledColor; float SampledAlpha;
    RayHitResult() : Distance(0.0f), SampledColor(1.0f, 1.0f, 1.0f), SampledAlpha(1.0f) {}
};
class RRay {
public:
    RVec3 Origin, Direction; float Distance;
    RRay() : Distance(0.0f) {}
    RRay(const RVec3& o, const RVec3& d, float dist) : Origin(o), Direction(d), Distance(dist) {}
};
struct RenderOption { bool UseBaseColor; RenderOption() : UseBaseColor(false) {} };

// ---- Src/Shapes.h / Src/MeshShape.h ----------------------------------------------------------------------------------------
class RShape {
public:
    virtual ~RShape() {}
    virtual int AddTo(rtw_scene* scene) const = 0;
};
class RMeshShape : public RShape {
public:
    explicit RMeshShape(const std::string& Filename) : Path(Filename) {}
    static std::unique_ptr<RMeshShape> Create(const std::string& Filename) { return std::unique_ptr<RMeshShape>(new RMeshShape(Filename)); }
    int AddTo(rtw_scene* scene) const override { int idx = -1; RtwCheck(rtw_scene_add_mesh_obj(scene, Path.c_str(), &idx)); return idx; }
private:
    std::string Path;
};

// RSphere / RPlane / RCapsule (Src/Shapes.h:46-112)
class RSphere : public RShape {
public:
    RVec3 Center; float Radius;
    RSphere(const RVec3& InCenter, float InRadius) : Center(InCenter), Radius(InRadius) {}
    static std::unique_ptr<RSphere> Create(const RVec3& InCenter, float InRadius) { return std::unique_ptr<RSphere>(new RSphere(InCenter, InRadius)); }
    int AddTo(rtw_scene* scene) const override { int idx = -1; const float c[3] = { Center.x, Center.y, Center.z }; RtwCheck(rtw_scene_add_sphere(scene, c, Radius, &idx)); return idx; }
};
class RPlane : public RShape {
public:
    RVec3 Normal, Point;
    RPlane(const RVec3& InNormal, const RVec3& InPoint) : Normal(InNormal), Point(InPoint) {}
    static std::unique_ptr<RShape> Create(const RVec3& InNormal, const RVec3& InPoint) { return std::unique_ptr<RShape>(new RPlane(InNormal, InPoint)); }
    int AddTo(rtw_scene* scene) const override
    {
        int idx = -1; const float n[3] = { Normal.x, Normal.y, Normal.z }, q[3] = { Point.x, Point.y, Point.z };
        RtwCheck(rtw_scene_add_plane(scene, n, q, &idx)); return idx;
    }
};
class RCapsule : public RShape {
public:
    RVec3 Start, End; float Radius;
    RCapsule(const RVec3& InStart, const RVec3& InEnd, float InRadius) : Start(InStart), End(InEnd), Radius(InRadius) {}
    static std::unique_ptr<RShape> Create(const RVec3& InStart, const RVec3& InEnd, float InRadius) { return std::unique_ptr<RShape>(new RCapsule(InStart, InEnd, InRadius)); }
    int AddTo(rtw_scene* scene) const override
    {
        int idx = -1; const float a[3] = { Start.x, Start.y, Start.z }, b[3] = { End.x, End.y, End.z };
        RtwCheck(rtw_scene_add_capsule(scene, a, b, Radius, &idx)); return idx;
    }
};

class RTriangle : public RShape {
public:
    RVec3 Points[3];
    RTriangle(const RVec3& p0, const RVec3& p1, const RVec3& p2) { Points[0] = p0; Points[1] = p1; Points[2] = p2; }
    static std::unique_ptr<RShape> Create(const RVec3& p0, const RVec3& p1, const RVec3& p2) { return std::unique_ptr<RShape>(new RTriangle(p0, p1, p2)); }
    int AddTo(rtw_scene* scene) const override
    {
        int idx = -1; float p[3][3];
        for (int i = 0; i < 3; i++) { p[i][0] = Points[i].x; p[i][1] = Points[i].y; p[i][2] = Points[i].z; }
        RtwCheck(rtw_scene_add_triangle(scene, p[0], p[1], p[2], &idx)); return idx;
    }
};

// ---- device context + frame buffers (accuBuffer[] / bitcolor[], Src/RayTracerProgram.cpp:49,77) ----------------------------------
class RtwDevice {
public:
    explicit RtwDevice(int device = 0) : Ctx(nullptr) { RtwCheck(rtw_context_create(device, &Ctx)); }
    ~RtwDevice() { rtw_context_destroy(Ctx); }
    rtw_context* Get() const { return Ctx; }
    void Synchronize() { RtwCheck(rtw_context_synchronize(Ctx)); }
private:
    RtwDevice(const RtwDevice&); RtwDevice& operator=(const RtwDevice&);
    rtw_context* Ctx;
};
class ColorBuffer {
public:
    ColorBuffer(RtwDevice& dev, int width, int height) : Fb(nullptr), Width(width), Height(height) { RtwCheck(rtw_framebuffer_create(dev.Get(), width, height, &Fb)); }
    ~ColorBuffer() { rtw_framebuffer_destroy(Fb); }
    rtw_framebuffer* Get() const { return Fb; }
    int bitmapWidth() const { return Width; }
    int bitmapHeight() const { return Height; }
    std::vector<Pixel> bitcolor() { std::vector<Pixel> p((size_t)Width * Height); RtwCheck(rtw_framebuffer_resolve_argb(Fb, p.data())); return p; }
    std::vector<float> accuBuffer() { std::vector<float> a((size_t)Width * Height * 4); RtwCheck(rtw_framebuffer_read_float(Fb, a.data())); return a; }
private:
    ColorBuffer(const ColorBuffer&); ColorBuffer& operator=(const ColorBuffer&);
    rtw_framebuffer* Fb; int Width, Height;
};

// ---- Src/RayTracerScene.h --------------------------------------------------------------------------------------------------------
class RayTracerScene {
public:
    explicit RayTracerScene(RtwDevice& dev) : Scene(nullptr), Committed(false) { RtwCheck(rtw_scene_create(dev.Get(), &Scene)); }
    ~RayTracerScene() { rtw_scene_destroy(Scene); }
    void AddShape(std::unique_ptr<RShape> Shape, std::unique_ptr<ISurfaceMaterial> SurfaceMaterial)
    {
        const int idx = Shape->AddTo(Scene);
        if (SurfaceMaterial) {
            std::vector<rtw_material_node> nodes;
            SurfaceMaterial->Flatten(nodes);
            RtwCheck(rtw_scene_set_material(Scene, idx, nodes.data(), (int)nodes.size()));
        }
    }
    // closest hit of one ray; returns the shape index or -1 (Src/RayTracerScene.cpp:99-125)
    int FindIntersectionWithScene(RRay TestRay, RayHitResult& OutResult)
    {
        Commit();
        const float ray[7] = { TestRay.Origin.x, TestRay.Origin.y, TestRay.Origin.z, TestRay.Direction.x, TestRay.Direction.y, TestRay.Direction.z, TestRay.Distance };
        float h[11]; int32_t shape = -1, tri = -1;
        RtwCheck(rtw_trace_closest(Scene, ray, 1, h, &shape, &tri));
        if (shape >= 0) {
            OutResult.HitPosition = RVec3(h[0], h[1], h[2]); OutResult.HitNormal = RVec3(h[3], h[4], h[5]); OutResult.Distance = h[6];
            OutResult.SampledColor = RVec3(h[7], h[8], h[9]); OutResult.SampledAlpha = h[10];
        }
        return shape;
    }
    // radiance along one ray (Src/RayTracerScene.cpp:31-97); (pixel, sample) select the random stream
    RVec3 RayTrace(const RRay& InRay, int MaxBounceTimes, const RenderOption& InOption, uint32_t pixel, uint32_t sample, uint32_t seed, int width, int height)
    {
        Commit();
        const float ray[7] = { InRay.Origin.x, InRay.Origin.y, InRay.Origin.z, InRay.Direction.x, InRay.Direction.y, InRay.Direction.z, InRay.Distance };
        const uint32_t key[2] = { pixel, sample };
        float rgb[3];
        RtwCheck(rtw_ray_trace(Scene, ray, key, 1, MaxBounceTimes, InOption.UseBaseColor ? 1 : 0, seed, width, height, rgb));
        return RVec3(rgb[0], rgb[1], rgb[2]);
    }
    void Commit() { if (!Committed) { RtwCheck(rtw_scene_commit(Scene)); Committed = true; } }
    rtw_scene* Get() { Commit(); return Scene; }
private:
    RayTracerScene(const RayTracerScene&); RayTracerScene& operator=(const RayTracerScene&);
    rtw_scene* Scene; bool Committed;
};

// ---- Src/RayTracerProgram.cpp:131 ---------------------------------------------------------------------------------------------------
// One pass over pixels begin..end (inclusive).  PassIndex / Seed pick the random streams (the reference draws from rand()).
inline void ThreadWorker_Render(RayTracerScene& Scene, ColorBuffer& Buffer, int begin, int end, int MaxBounceCount,
                                const RenderOption& InOption = RenderOption(), int PassIndex = 0, uint32_t Seed = 12345, int SubSamples = 4)
{
    RtwCheck(rtw_render_range(Scene.Get(), Buffer.Get(), begin, end, MaxBounceCount, InOption.UseBaseColor ? 1 : 0, PassIndex, SubSamples, Seed));
}

// ---- Src/Texture.h ------------------------------------------------------------------------------------------------------------------
class RTexture {
public:
    static bool SaveBufferToPNG(const std::string& Filename, const UINT32* Pixels, int width, int height)
    {
        return rtw_png_save_argb(Filename.c_str(), Pixels, width, height) == RTW_OK;
    }
};

// ---- Src/Linux/RenderWindow_X11.h:11-30 (and its Windows / OSX twins): the display hook ------------------------------------------------
// The reference blits bitcolor[] to a native window and writes the progress line into its title.  On a headless GPU node the same
// four calls feed a SINK: every presented frame (the resolved 0xAARRGGBB image) and every title go to callbacks, if set, and are
// counted.  SetRenderBufferParameters names the caller's pixel buffer exactly like the reference's call (width, height, buffer);
// the renderer copies the device image into it before each Present().  A sink that can take the image where it lies -- a device pointer (an interop
// surface, an encoder, a kernel of the caller's) -- sets a DEVICE sink instead: it is called with the device address of the 0xAARRGGBB image after every
// update, the renders that produced it already complete, and no host copy is made at all (8.3 MB per present at 1080p otherwise).
class RenderWindow {
public:
    typedef void (*FrameSink)(void* User, const Pixel* Pixels, int Width, int Height);
    typedef void (*DeviceFrameSink)(void* User, const void* DevicePixels, int Width, int Height);
    typedef void (*TitleSink)(void* User, const char* Title);
    RenderWindow() : Width(0), Height(0), Buffer(nullptr), OnFrame(nullptr), OnDeviceFrame(nullptr), OnTitle(nullptr), User(nullptr), Frames(0), Created(false), Closing(false) {}
    bool Create(int InWidth, int InHeight) { if (InWidth <= 0 || InHeight <= 0) return false; Width = InWidth; Height = InHeight; Created = true; return true; }
    void SetRenderBufferParameters(int BufferWidth, int BufferHeight, void* InBuffer) { Width = BufferWidth; Height = BufferHeight; Buffer = (Pixel*)InBuffer; }
    void SetTitle(const char* InTitle) { Title = InTitle ? InTitle : ""; if (OnTitle) OnTitle(User, Title.c_str()); }
    // the reference's loop pumps native events until the window closes; a sink has no events: it returns once the renderer is done
    void RunWindowLoop() { Closing = true; }
    void SetSinks(FrameSink InFrame, TitleSink InTitle, void* InUser) { OnFrame = InFrame; OnTitle = InTitle; User = InUser; }
    void SetDeviceSink(DeviceFrameSink InFrame, void* InUser) { OnDeviceFrame = InFrame; User = InUser; }
    bool WantsDeviceFrames() const { return OnDeviceFrame != nullptr; }
    // called by the renderer after the buffer named in SetRenderBufferParameters has been refreshed
    void Present() { Frames++; if (OnFrame && Buffer) OnFrame(User, Buffer, Width, Height); }
    // ... or with the image still in device memory (no host buffer involved)
    void PresentDevice(const void* DevicePixels, int W, int H) { Frames++; if (OnDeviceFrame) OnDeviceFrame(User, DevicePixels, W, H); }
    Pixel* RenderBuffer() const { return Buffer; }
    int BufferWidth() const { return Width; }
    int BufferHeight() const { return Height; }
    const std::string& GetTitle() const { return Title; }
    int PresentedFrames() const { return Frames; }
    bool IsCreated() const { return Created; }
private:
    int Width, Height; Pixel* Buffer; FrameSink OnFrame; DeviceFrameSink OnDeviceFrame; TitleSink OnTitle; void* User; std::string Title; int Frames; bool Created, Closing;
};

// ---- Src/RayTracerProgram.cpp:242-268 ----------------------------------------------------------------------------------------------
// "12ms" below a second, else "3s", "2m:5s", "1h:0m:7s" (minutes and seconds are the remainders, as the reference computes them)
inline void FormatTimeString(char* Buffer, int BufferSize, int Milliseconds)
{
    if (Milliseconds < 1000) { std::snprintf(Buffer, (size_t)BufferSize, "%dms", Milliseconds); return; }
    const int Hours = Milliseconds / 3600000;
    const int Minutes = Milliseconds / 60000 - Hours * 60;
    const int Seconds = Milliseconds / 1000 - Minutes * 60 - Hours * 3600;
    if (Hours != 0) std::snprintf(Buffer, (size_t)BufferSize, "%dh:%dm:%ds", Hours, Minutes, Seconds);
    else if (Minutes != 0) std::snprintf(Buffer, (size_t)BufferSize, "%dm:%ds", Minutes, Seconds);
    else std::snprintf(Buffer, (size_t)BufferSize, "%ds", Seconds);
}

// ---- Src/RayTracerProgram.cpp:270-422: the progressive render -------------------------------------------------------------------
// What the reference's UpdateBitmapPixels does, on the device: one base-colour preview pass, then TotalSamplesNum accumulated passes of
// the whole frame, a progress line after every update ("RayTracer - S: [n/N] | T: [elapsed / remaining] | F: [frame ms]", also the
// window title), cooperative quit, and the image saved as Output_<N>spp_<date>.png in the first of SavedImages/, ../SavedImages/,
// ../../SavedImages/ that holds an Output.txt.
struct RtwProgressive {
    int TotalSamplesNum;        // 500 in the reference
    int MaxBounceTimes;         // 10 in the reference
    int PassesPerUpdate;        // passes rendered between two progress lines / presents: 1 = the reference's rhythm; more lets the library
                                // batch the passes of an update into shared launches (several times faster on small frames)
    uint32_t Seed;
    const volatile bool* bQuit; // polled after every update (RayTracerProgram::IsTerminating); with several ranks rank 0's flag decides for all of them
    RenderWindow* Window;       // display hook, may be null
    int Rank, World;            // this process renders the 10-row tasks t with t % World == Rank ...
    rtw_comm* Comm;             // ... and the rows travel to rank 0 over RCCL before every present and before the image is saved (null with World == 1)
    bool ArgbOnlyGather;        // gather the displayable image only (4 B / pixel)
    bool Quiet;                 // no progress lines on stdout (the title still gets them)
    RtwProgressive() : TotalSamplesNum(500), MaxBounceTimes(10), PassesPerUpdate(1), Seed(12345), bQuit(nullptr), Window(nullptr), Rank(0), World(1), Comm(nullptr),
                       ArgbOnlyGather(false), Quiet(false) {}
};

// Returns the path of the image written; "" when no output folder was found, nothing was rendered, or this is not rank 0.
inline std::string UpdateBitmapPixels(RtwDevice& Device, RayTracerScene& Scene, ColorBuffer& Buffer, const RtwProgressive& Run)
{
    using Clock = std::chrono::system_clock;
    const int TaskRows = 10;                       // NumTaskRows
    const bool Root = Run.Rank == 0;
    if (Run.World > 1 && !Run.Comm) throw std::invalid_argument("UpdateBitmapPixels: several ranks need a communicator (RtwProgressive::Comm): without the gather rank 0 would show and save its own rows only");
    auto Show = [&](const char* Text) {            // gather (several ranks), refresh the window's buffer, present
        if (Run.World > 1) RtwCheck(rtw_gather_rows(Run.Comm, Buffer.Get(), TaskRows, Run.ArgbOnlyGather ? RTW_GATHER_ARGB : RTW_GATHER_ALL));
        if (!Root || !Run.Window) return;
        if (Text) Run.Window->SetTitle(Text);
        if (Run.Window->WantsDeviceFrames()) {     // the image where it lies: no host copy
            void* DeviceArgb = nullptr;
            RtwCheck(rtw_framebuffer_device_pointers(Buffer.Get(), nullptr, &DeviceArgb));
            Device.Synchronize();
            Run.Window->PresentDevice(DeviceArgb, Buffer.bitmapWidth(), Buffer.bitmapHeight());
        } else if (Run.Window->RenderBuffer() && Run.Window->BufferWidth() == Buffer.bitmapWidth() && Run.Window->BufferHeight() == Buffer.bitmapHeight()) {
            RtwCheck(rtw_framebuffer_resolve_argb(Buffer.Get(), Run.Window->RenderBuffer()));
            Run.Window->Present();
        }
    };
    // the preview: every material's base colour, no bounces
    RtwCheck(rtw_render_tasks(Scene.Get(), Buffer.Get(), TaskRows, Run.Rank, Run.World, Run.MaxBounceTimes, 1, 0, 4, Run.Seed));
    Show(nullptr);
    Device.Synchronize();
    const Clock::time_point Begin = Clock::now();
    Clock::time_point Previous = Begin;
    const int Step = Run.PassesPerUpdate > 0 ? Run.PassesPerUpdate : 1;
    RtwCheck(rtw_render_reserve(Scene.Get(), Buffer.Get(), TaskRows, Run.Rank, Run.World, Run.MaxBounceTimes, Step, 4));      // the updates' workspace, once, outside the loop
    int Rendered = 0;
    while (Rendered < Run.TotalSamplesNum) {
        const int Count = Run.TotalSamplesNum - Rendered < Step ? Run.TotalSamplesNum - Rendered : Step;
        RtwCheck(rtw_render_passes(Scene.Get(), Buffer.Get(), TaskRows, Run.Rank, Run.World, Run.MaxBounceTimes, 0, Rendered, Count, 4, Run.Seed));
        Rendered += Count;
        Device.Synchronize();
        const Clock::time_point Now = Clock::now();
        const long long SinceBegin = std::chrono::duration_cast<std::chrono::milliseconds>(Now - Begin).count();
        const long long Left = SinceBegin / Rendered * (Run.TotalSamplesNum - Rendered);           // integer milliseconds, as the reference's duration arithmetic
        const int SinceUpdate = (int)std::chrono::duration_cast<std::chrono::milliseconds>(Now - Previous).count();
        Previous = Now;
        char Spent[64], Remaining[64], Line[256];
        FormatTimeString(Spent, (int)sizeof Spent, (int)SinceBegin);
        FormatTimeString(Remaining, (int)sizeof Remaining, (int)Left);
        std::snprintf(Line, sizeof Line, "RayTracer - S: [%d/%d] | T: [%s / %s] | F: [%dms]", Rendered, Run.TotalSamplesNum, Spent, Remaining, SinceUpdate);
        if (Root && !Run.Quiet) std::printf("%s\n", Line);
        Show(Line);
        int Quit = (Run.bQuit && *Run.bQuit) ? 1 : 0;
        if (Run.World > 1) RtwCheck(rtw_comm_broadcast_int(Run.Comm, &Quit));      // rank 0 decides for every rank: one that left alone would leave the others in the next gather
        if (Quit) break;
    }
    // (every update's Show() has gathered, the last one after the last pass: rank 0 holds the whole image here.  The gather is a collective of the
    // ranks: it must never depend on something only one rank has, such as a window)
    Device.Synchronize();
    if (Root && !Run.Quiet) std::printf("Finished rendering image.\n");
    if (!Root || Rendered == 0) return std::string();
    // Output_<N>spp_<date>.png beside the marker file Output.txt
    char Stamp[80];
    const time_t Raw = time(nullptr);
    strftime(Stamp, sizeof Stamp, "%Y-%m-%d_%H-%M-%S", localtime(&Raw));
    std::string Folder;
    const char* Candidates[3] = { "SavedImages/", "../SavedImages/", "../../SavedImages/" };
    for (int i = 0; i < 3 && Folder.empty(); i++) {
        std::FILE* Marker = std::fopen((std::string(Candidates[i]) + "Output.txt").c_str(), "rb");
        if (Marker) { std::fclose(Marker); Folder = Candidates[i]; }
    }
    if (Folder.empty()) { std::printf("Unable to find the output folder SavedImages!\n"); return std::string(); }
    const std::string Filename = Folder + "Output_" + std::to_string(Run.TotalSamplesNum) + "spp_" + Stamp + ".png";
    const std::vector<Pixel> Image = Buffer.bitcolor();
    if (!RTexture::SaveBufferToPNG(Filename, Image.data(), Buffer.bitmapWidth(), Buffer.bitmapHeight())) return std::string();
    std::printf("Image saved as %s\n", Filename.c_str());
    return Filename;
}

// the reference's argument-less call with the sample count and bounce limit it compiles in
inline std::string UpdateBitmapPixels(RtwDevice& Device, RayTracerScene& Scene, ColorBuffer& Buffer, int TotalSamplesNum = 500, int MaxBounceTimes = 10,
                                      const volatile bool* bQuit = nullptr, uint32_t Seed = 12345, bool LogEveryPass = true)
{
    RtwProgressive Run;
    Run.TotalSamplesNum = TotalSamplesNum; Run.MaxBounceTimes = MaxBounceTimes; Run.bQuit = bQuit; Run.Seed = Seed;
    Run.PassesPerUpdate = LogEveryPass ? 1 : (TotalSamplesNum > 0 ? TotalSamplesNum : 1);
    return UpdateBitmapPixels(Device, Scene, Buffer, Run);
}
