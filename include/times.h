// Timing accumulator and `-m time` report of the SPH driver.
//
// Same type name, field names/order and byte-for-byte the same table as the
// reference's src/times.h:5-35, so main.cpp:68-76-style callers work unchanged:
//
//   Operation            Per frame       Total
//   ---------------------------------------------
//   Grid construction    0.00123        0.12346
//   SPH update           0.01500        1.50000
//   Data transfer        0.00042        0.04210
//
// Grid construction = cell hash + radix sort + gather/cell ranges (+ table
// clear); SPH update = density + force/integrate; Data transfer = the part of
// the device->host position copy the pipeline could not overlap.
#ifndef SPH_TIMES_H
#define SPH_TIMES_H

#include <cstdio>
#include <iostream>

struct Times {
    double buildGrid = 0.0;
    double sphUpdate = 0.0;
    double memcpy = 0.0;
    int iters = 0;
};

inline void displayTimes(Times *times) {
    const double n = times->iters ? (double)times->iters : 1.0;
    const double perGrid = times->iters ? times->buildGrid / n : 0.0;
    const double perSph = times->iters ? times->sphUpdate / n : 0.0;
    const double perCopy = times->iters ? times->memcpy / n : 0.0;
    char line[5][96];
    std::snprintf(line[0], sizeof line[0], "%-12s%18s%12s", "Operation", "Per frame", "Total");
    std::snprintf(line[1], sizeof line[1], "%s", "---------------------------------------------");
    std::snprintf(line[2], sizeof line[2], "%-11s%11.5f%15.5f", "Grid construction", perGrid, times->buildGrid);
    std::snprintf(line[3], sizeof line[3], "%-12s%16.5f%15.5f", "SPH update", perSph, times->sphUpdate);
    std::snprintf(line[4], sizeof line[4], "%-12s%15.5f%15.5f", "Data transfer", perCopy, times->memcpy);
    for (auto &l : line) std::cout << l << std::endl;
}

#endif
