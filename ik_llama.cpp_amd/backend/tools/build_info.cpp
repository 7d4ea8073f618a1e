// What the reference's cmake generates from common/build-info.cpp.in (the four build-identification globals that common.cpp and
// llama-bench.cpp print): Makefile.llama does not run the reference's build system, so it supplies them here when the reference tree
// holds no generated common/build-info.cpp.
int LLAMA_BUILD_NUMBER = 0;
char const *LLAMA_COMMIT = "unknown";
char const *LLAMA_COMPILER = "g++ (Makefile.llama)";
char const *LLAMA_BUILD_TARGET = "x86_64-linux-gnu";
