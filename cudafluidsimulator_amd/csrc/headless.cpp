// Stand-in for the GLUT front end when ./sph is built without display.cpp.
// It defines the two globals the simulator links against (display.cpp:19-20)
// and a startVisualization() that steps the simulation without a window.
// Never linked together with display.cpp (duplicate symbols by design).
#include <cstdio>
#include <cstdlib>

#include "sha256.h"
#include "simulator.h"

bool mouseClicked = false;
int2 clickCoords;

extern "C" void glutInit(int *, char **) {}

void startVisualization(Simulator *simulator) {
    int frames = 100;
    if (const char *e = getenv("SPH_FREE_FRAMES")) frames = atoi(e);
    fprintf(stderr, "sph: built without GLUT -- running %d frames headless\n", frames);
    for (int f = 0; f < frames; ++f) {
        if (f == frames / 2 && getenv("SPH_FREE_CLICK")) {
            mouseClicked = true;
            clickCoords = make_int2(400, 300);
        }
        simulator->simulate();
    }
    const float3 *p = simulator->getPosition();
    if (p && simulator->settings->numParticles > 0)
        printf("particle 0 after %d frames: (%f, %f, %f)\n", frames, p[0].x, p[0].y, p[0].z);
    if (p && getenv("SPH_PRINT_SHA256"))
        printf("positions_sha256 %s\n",
               sha256_hex(p, (size_t)simulator->settings->numParticles * sizeof(float3)).c_str());
}
