// Finds and loads cardinality packs (cardinality_pack.h).
#include "cardinality_pack.h"

#include <dlfcn.h>

#include <cstdlib>
#include <map>
#include <mutex>
#include <string>

namespace bn {

namespace {

std::string library_directory() {
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&cardinality_pack_directory), &info) == 0 || !info.dli_fname) return ".";
  const std::string path = info.dli_fname;
  const size_t cut = path.find_last_of('/');
  return cut == std::string::npos ? "." : path.substr(0, cut);
}

const CardinalityPack* load_pack(int32_t d0) {
  if (getenv("BN_NO_PACKS")) return nullptr;
  const std::string path = library_directory() + "/libbn_hip_d" + std::to_string(d0) + ".so";
  void* handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!handle) return nullptr;
  using Entry = const CardinalityPack* (*)();
  Entry entry = reinterpret_cast<Entry>(dlsym(handle, "bn_cardinality_pack"));
  const CardinalityPack* pack = entry ? entry() : nullptr;
  if (!pack || pack->abi != expected_pack_abi() || pack->d0 != d0) {
    dlclose(handle);  // built for another version of the library
    return nullptr;
  }
  return pack;
}

const ShapePack* load_shape_pack(int32_t d0, int32_t d1, int32_t s) {
  if (getenv("BN_NO_PACKS")) return nullptr;
  const std::string path = library_directory() + "/libbn_hip_p" + std::to_string(d0) + "x" + std::to_string(d1) + "x" +
                           std::to_string(s) + ".so";
  void* handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!handle) return nullptr;
  using Entry = const ShapePack* (*)();
  Entry entry = reinterpret_cast<Entry>(dlsym(handle, "bn_shape_pack"));
  const ShapePack* pack = entry ? entry() : nullptr;
  if (!pack || pack->abi != expected_pack_abi() || pack->d0 != d0 || pack->d1 != d1 || pack->s != s) {
    dlclose(handle);
    return nullptr;
  }
  return pack;
}

}  // namespace

const ShapePack* shape_pack(int32_t d0, int32_t d1, int32_t s) {
  static std::mutex lock;
  static std::map<int64_t, const ShapePack*> known;
  if (d0 < 2 || d1 < 2 || s < 2 || d0 > 64 || d1 > 64 || s > 64) return nullptr;
  const int64_t key = ((int64_t)d0 << 32) | ((int64_t)d1 << 16) | s;
  std::lock_guard<std::mutex> guard(lock);
  auto it = known.find(key);
  if (it != known.end()) return it->second;
  const ShapePack* pack = load_shape_pack(d0, d1, s);  // (a miss is not remembered, as for cardinality packs)
  if (pack) known.emplace(key, pack);
  return pack;
}

const char* cardinality_pack_directory() {
  static const std::string dir = library_directory();
  return dir.c_str();
}

const CardinalityPack* cardinality_pack(int32_t d0) {
  static std::mutex lock;
  static std::map<int32_t, const CardinalityPack*> known;
  if (d0 < 2 || d0 > 64) return nullptr;
  std::lock_guard<std::mutex> guard(lock);
  auto it = known.find(d0);
  if (it != known.end()) return it->second;
  // (a miss is not remembered: the pack may be built while the process runs; plans ask when they are compiled)
  const CardinalityPack* pack = load_pack(d0);
  if (pack) known.emplace(d0, pack);
  return pack;
}

}  // namespace bn
