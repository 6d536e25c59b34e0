// Public C++ surface of the MI355X SPH simulator.
//
// Field-for-field / method-for-method the same surface as the reference's
// src/simulator.h:6-74 so its front ends (src/main.cpp, src/display.cpp) build
// against this header unchanged: the physics and click-box macros, `Settings`
// (positional brace-init in main.cpp:62-63 depends on the field order),
// `Particle`, and `class Simulator`.  What differs is private: the reference
// keeps CUDA device pointers here, this class keeps one opaque handle of the
// C-ABI in sph_c_api.h, behind which the hand-written gfx950 kernels live.
#ifndef SPH_SIMULATOR_H
#define SPH_SIMULATOR_H

#include <stdio.h>

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_vector_types.h> // float3, int2, make_int2 (types only)

#include "times.h"

#define PI 3.14159265f
#define MASS 0.02f
#define GAS_CONSTANT 1.f
#define REST_DENSITY 1000.f
#define VISCOSITY 1.f
#define GRAVITY -9.8f
#define ELASTICITY 0.5f

// window-pixel box inside which a left click pushes the fluid (display.cpp:22-32)
#define BOX_MAX_X (600)
#define BOX_MIN_X (200)
#define BOX_MAX_Y (450)
#define BOX_MIN_Y (150)

struct Settings {
    bool randomInit;
    int numParticles;
    float h;

    // Pre-computed constants
    float v_kernel_coeff;
    float d_kernel_coeff;

    float boxDim;
    float numCellsPerDim;
    float timestep;
};

// Legacy per-particle record of the reference's AoS layout.  The simulator no
// longer stores particles this way (state is two key-sorted float4 streams on
// the device); the type is kept so code that names it still compiles.
struct Particle {
    float3 position, velocity, force;
    float density, pressure;
    struct Particle *next;

    Particle(float3 pos)
        : position(pos), velocity{0.f, 0.f, 0.f}, force{0.f, 0.f, 0.f}, density(0.f),
          pressure(0.f), next(NULL) {}

    void display() { printf("(%f, %f, %f)\n", position.x, position.y, position.z); }
};

struct sph_handle;
struct sph_mgpu;

class Simulator {
  private:
    struct sph_handle *impl; // one GPU (include/sph_c_api.h)
    struct sph_mgpu *multi;  // SPH_GPUS=N: z-slabs over N GPUs (include/sph_mgpu.h)

  public:
    const Settings *settings;

    Simulator(Settings *settings);
    virtual ~Simulator();

    void setup();

    const float3 *getPosition();

    void simulate();
    void simulateAndTime(Times *times);
    void moveParticles(int2 mouse_pos);
};

#endif
