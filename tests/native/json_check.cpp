#include <cstdio>
#include <fstream>
#include <iostream>
#include <string>
#include "json_min.hpp"
#include "scene.hpp"
int main(int argc, char** argv) {
    // stdin: one document per line -> parse + dump(4) + dump(-1); then load scene files given as args
    std::string line; size_t ok = 0, bad = 0, bytes = 0;
    while (std::getline(std::cin, line)) {
        try { auto j = srt_host::Json::parse(line); bytes += j.dump(4).size() + j.dump(-1).size(); ++ok; } catch (const std::exception&) { ++bad; }
    }
    for (int i = 1; i < argc; ++i) { srt_host::Scene s(argv[i]); s.Load(); bytes += s.Dump().size(); (void)s.Flatten(); (void)s.MeshViews(); }
    std::printf("ok %zu bad %zu bytes %zu\n", ok, bad, bytes);
}
