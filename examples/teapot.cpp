// teapot.cpp -- stand-in for the reference's `fn main` (src/main.rs:381-523) without the window:
// builds the default scene through the C++ mirror of the reference API and runs the redraw loop of
// src/main.rs:466-501 on the GPU for a number of frames: the sun moves with the time since start,
// a line "fps: ..." is printed once a second, and the last frame is written as a PPM where the
// reference presents it in a window.
//   g++ -O2 -std=c++17 examples/teapot.cpp -Iraytracing_amd/host -Lraytracing_amd/lib -lrays_host -lrays_hip \
//       -Wl,-rpath,$PWD/raytracing_amd/lib -o teapot && ./teapot tests/golden/teapot.obj out.ppm [frames]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "rays.hpp"

int main(int argc, char** argv) {
    const char* obj = argc > 1 ? argv[1] : "teapot.obj";
    const char* out = argc > 2 ? argv[2] : "frame.ppm";
    const long n_frames = argc > 3 ? std::atol(argv[3]) : 1;
    using clock = std::chrono::steady_clock;
    try {
        rays::World world(0);
        rays::default_scene(obj, &world);
        world.pack();
        std::printf("len: %u\n", world.num_tris());  // src/main.rs:456-459
        rays::FrameParams fp;  // 800 x 800, the reference's constants
        std::vector<uint8_t> frame;
        const clock::time_point start = clock::now();
        clock::time_point frame_timer = start;
        long frames = 0;
        for (long i = 0; i < n_frames; ++i) {
            // With one frame asked for, the sun stands where the reference's first redraw puts it.
            const float t = n_frames > 1 ? std::chrono::duration<float>(clock::now() - start).count() : 0.0f;
            world.sun = rays::sun_at(t);
            fp.seed = (uint32_t)i;  // fresh scatter directions per frame, as rand::random gives the reference
            world.draw(frame, fp);
            ++frames;
            const float since = std::chrono::duration<float>(clock::now() - frame_timer).count();
            if (since >= 1.0f) {  // src/main.rs:488-496
                std::printf("fps: %f\n", (float)frames / since);
                frames = 0;
                frame_timer = clock::now();
            }
        }
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
