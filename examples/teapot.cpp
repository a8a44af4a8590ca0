// teapot.cpp -- stand-in for the reference's `fn main` (src/main.rs:381-464) without the window:
// builds the default scene through the C++ mirror of the reference API, renders one frame on the
// GPU and writes it as a PPM.
//   g++ -O2 -std=c++17 examples/teapot.cpp -Iraytracing_amd/host -Lraytracing_amd/lib -lrays_host -lrays_hip \
//       -Wl,-rpath,$PWD/raytracing_amd/lib -o teapot && ./teapot tests/golden/teapot.obj out.ppm
#include <cmath>
#include <cstdio>
#include <vector>

#include "rays.hpp"

int main(int argc, char** argv) {
    const char* obj = argc > 1 ? argv[1] : "teapot.obj";
    const char* out = argc > 2 ? argv[2] : "frame.ppm";
    try {
        rays::World world(0);
        rays::default_scene(obj, &world);
        world.pack();
        std::printf("len: %u\n", world.num_tris());  // src/main.rs:456-459
        const float t = 0.0f;                        // src/main.rs:479-483 at t = 0
        world.sun = rays::Vec3{std::sin(t / 10.0f) * 100000.0f, 25000.0f, std::cos(t / 10.0f) * 100000.0f};
        rays::FrameParams fp;  // 800 x 800, the reference's constants
        std::vector<uint8_t> frame;
        world.draw(frame, fp);
        FILE* f = std::fopen(out, "wb");
        if (!f) return 1;
        std::fprintf(f, "P6\n%u %u\n255\n", fp.width, fp.height);
        for (size_t i = 0; i < (size_t)fp.width * fp.height; ++i) std::fwrite(&frame[4 * i], 1, 3, f);
        std::fclose(f);
    } catch (const rays::Error& e) {
        std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
