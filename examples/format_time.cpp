// format_time.cpp -- prints the facade's FormatTimeString (Src/RayTracerProgram.cpp:242-268) for the millisecond values given (no GPU needed)
#include <cstdio>
#include <cstdlib>

#include "RayTracerWin.hpp"

int main(int argc, char** argv)
{
    for (int i = 1; i < argc; i++) {
        char buf[256];
        FormatTimeString(buf, (int)sizeof buf, std::atoi(argv[i]));
        std::printf("%s %s\n", argv[i], buf);
    }
    return 0;
}
