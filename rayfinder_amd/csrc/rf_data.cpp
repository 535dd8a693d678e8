#include "rf_data.hpp"

// The assembler resolves the file names through -Wa,-I<repo>/rayfinder_amd/data.
__asm__(".section .rodata\n"
        ".balign 16\n"
        "rf_embedded_hw_sky_tables:\n"
        ".incbin \"hw_sky_tables.bin\"\n"
        ".balign 16\n"
        "rf_embedded_blue_noise:\n"
        ".incbin \"blue_noise_128x128_rg8.bin\"\n"
        ".previous\n");
extern "C" const float   rf_embedded_hw_sky_tables[];
extern "C" const uint8_t rf_embedded_blue_noise[];

namespace rf
{
const float*   hwSkyTables() { return rf_embedded_hw_sky_tables; }
const uint8_t* blueNoiseTable() { return rf_embedded_blue_noise; }
} // namespace rf
