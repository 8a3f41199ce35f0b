// make_cpp_archive.cpp — writes a weight archive the way the reference's C++ front end does, for the converter's tests
// (SURVEY.md §8(f) rank 4).  The reference loads its weights with torch::load(model_, common::model_path)
// (/root/reference/orb_slam2/src/cv/sp_extractor.cpp:355) into a torch::nn::Module whose twelve Conv2d children are
// registered as conv1a ... convDb (:46-62), so a real `superpoint.pt` for it is what torch::save(module, path) of such a
// module produces: a C++-frontend (OutputArchive) file, not a Python torch.jit.script / state_dict one.
// This program is the builder's own (not reference code): same child names and kernel sizes, channel counts divided by
// `div`, values from an integer hash so that the test can recompute them without libtorch.
//   build (dev container only; pip libtorch):  see tools/cpp_archive/build.sh
//   run:  make_cpp_archive <out.pt> <div>
#include <torch/torch.h>

#include <cstdint>
#include <cstdlib>
#include <iostream>

struct Front : torch::nn::Module {
  Front(int div) {
    const char *names[12] = {"conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "convDa", "convDb"};
    const int cin[12] = {1, 64, 64, 64, 64, 128, 128, 128, 128, 256, 128, 256};
    const int cout[12] = {64, 64, 64, 64, 128, 128, 128, 128, 256, 65, 256, 256};
    const int ks[12] = {3, 3, 3, 3, 3, 3, 3, 3, 3, 1, 3, 1};
    for (int l = 0; l < 12; ++l) {
      const int ci = l == 0 ? 1 : std::max(1, cin[l] / div), co = std::max(1, cout[l] / div);
      auto conv = register_module(names[l], torch::nn::Conv2d(torch::nn::Conv2dOptions(ci, co, ks[l]).stride(1).padding(ks[l] / 2)));
      torch::NoGradGuard g;
      float *w = conv->weight.data_ptr<float>();
      for (int64_t i = 0; i < conv->weight.numel(); ++i)
        w[i] = (float)((((uint32_t)i * 2654435761u + (uint32_t)l * 0x01000193u) >> 8) & 0xFFFFu) / 65536.0f - 0.5f;
      float *b = conv->bias.data_ptr<float>();
      for (int64_t i = 0; i < conv->bias.numel(); ++i)
        b[i] = (float)((((uint32_t)i * 40503u + (uint32_t)l * 13u + 7u) >> 4) & 0xFFFu) / 4096.0f - 0.5f;
    }
  }
};

int main(int argc, char **argv) {
  if (argc != 3) { std::cerr << "usage: make_cpp_archive <out.pt> <div>\n"; return 2; }
  auto m = std::make_shared<Front>(std::atoi(argv[2]));
  torch::save(m, argv[1]);
  return 0;
}
