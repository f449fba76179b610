// Counted, shared mappings of inter-process memory handles: one open per (handle, device) and process however many ranks (host threads) of the
// process ask for it, the last close unmaps.  The open / close calls themselves come from the policy class (dist.cpp: hipIpcOpenMemHandle /
// hipIpcCloseMemHandle; the sanitizer tier: fakes), so this header has no HIP in it.
#pragma once
#include <array>
#include <map>
#include <mutex>

namespace kr {

using SharedMappingKey = std::array<char, 68>;          // 64 bytes of handle + the device it is opened on

template <class Ops>                                    // Ops::open(void** ptr, const SharedMappingKey&) -> 0 on success; Ops::close(void* ptr)
class SharedMappings {
    struct Entry { void* ptr; int refs; };
    std::mutex mu_;
    std::map<SharedMappingKey, Entry> open_;
public:
    // Two threads asking for the same handle at the same moment: the second waits for the first one's open (the lock is held across it -- an
    // open is rare and short, and the runtime does not promise to survive two concurrent opens of one handle) and shares its mapping.
    int open(void** ptr, const SharedMappingKey& key) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = open_.find(key);
        if (it != open_.end()) { ++it->second.refs; *ptr = it->second.ptr; return 0; }
        const int e = Ops::open(ptr, key);
        if (e == 0) open_.emplace(key, Entry{*ptr, 1});
        return e;
    }
    void close(void* ptr) {
        std::lock_guard<std::mutex> g(mu_);
        for (auto it = open_.begin(); it != open_.end(); ++it)
            if (it->second.ptr == ptr) {
                if (--it->second.refs == 0) { Ops::close(ptr); open_.erase(it); }
                return;
            }
        Ops::close(ptr);                                // (not one of ours)
    }
    size_t size() { std::lock_guard<std::mutex> g(mu_); return open_.size(); }
};

}  // namespace kr
