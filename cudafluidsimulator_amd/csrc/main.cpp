// ./sph -- command-line driver with the reference's interface (src/main.cpp):
//   -n <NUM_PARTICLES>  -i <random/grid>  -m <free/time>  -?
// Same defaults (1000 / grid / time), same rejection of bad -i/-m values, same
// derived constants, 100 timed steps and the same table.  One extra,
// machine-readable line follows the table; with SPH_PRINT_SHA256 set, one more line
// holds the sha256 of the getPosition() array (numParticles x float3, particle-id
// order) so tests can compare this C++ path with the oracle's checksums.
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>

#include "sha256.h"
#include "simulator.h"

void startVisualization(Simulator *simulator);
extern "C" void glutInit(int *, char **);

static void usage() {
    printf("Program Options:\n");
    printf("  -n  <NUM_PARTICLES>    Number of particles to simulate\n");
    printf("  -i  <random/grid>      Initialization mode: random or grid\n");
    printf("  -m  <free/time>        Execution mode: free or timed\n");
    printf("  -?                     This message\n");
}

int main(int argc, char **argv) {
    int numParticles = 1000;
    bool randomInit = false;
    bool benchmark = true;

    int opt;
    while ((opt = getopt(argc, argv, "n:i:m:?")) != -1) {
        std::string arg = optarg ? optarg : "";
        if (opt == 'n') {
            numParticles = std::stoi(arg);
        } else if (opt == 'i') {
            if (arg != "random" && arg != "grid") {
                std::cout << "Invalid argument for option -i: " << arg << std::endl;
                usage();
                return 1;
            }
            randomInit = (arg == "random");
        } else if (opt == 'm') {
            if (arg != "time" && arg != "free") {
                std::cout << "Invalid argument for option -m: " << arg << std::endl;
                usage();
                return 1;
            }
            benchmark = (arg == "time");
        } else {
            usage();
            return 1;
        }
    }

    // same expressions and types as the reference (float h, pow in double)
    float h = .1f;
    float h_pow_6 = pow(h, 6);
    float h_pow_9 = pow(h, 9);
    float v_kernel_coeff = 45.f / (PI * h_pow_6);
    float d_kernel_coeff = 315.f / (64.f * PI * h_pow_9);
    Settings settings = {randomInit,     numParticles, h,   v_kernel_coeff,
                         d_kernel_coeff, 10.f,         100, .01};

    Simulator *simulator = new Simulator(&settings);
    simulator->setup();

    if (benchmark) {
        const int numIters = 100;
        Times times;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < numIters; i++) simulator->simulateAndTime(&times);
        const float3 *last = simulator->getPosition(); // last frame's positions have landed
        double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        displayTimes(&times);
        printf("{\"particle_steps_per_s\": %.6e, \"n\": %d, \"steps\": %d, \"wall_s\": %.6f}\n",
               (double)numParticles * numIters / wall, numParticles, numIters, wall);
        if (getenv("SPH_PRINT_SHA256") && last)
            printf("positions_sha256 %s\n", sha256_hex(last, (size_t)numParticles * sizeof(float3)).c_str());
    } else {
        glutInit(&argc, argv);
        startVisualization(simulator);
    }
    delete simulator;
    return 0;
}
