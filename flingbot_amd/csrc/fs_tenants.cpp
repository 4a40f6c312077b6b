// fs_tenants.cpp -- the co-tenant table: which live processes of this user drive the same physical GPU through this library.
//
// Why: the reference runs one PyFleX per Ray worker on ONE GPU (`--num_processes 16`, README.md:147-148, utils.py:144-157).
// A lone cloth steps fastest on the streaming kernels (129 small launches per frame spread over the chip), but the chip
// dispatches those launches at one rate IN TOTAL, so sixteen processes doing that share one process's rate -- while the fused
// kernel is one launch per frame on one compute unit, and sixteen of those do run side by side.  Which back-end
// is right therefore depends on how many processes share the device, and an unmodified caller of `pyflex` cannot tell us: the
// processes find each other here.
//
// Mechanism: one small file per (user, device) in /dev/shm (FLINGSIM_TENANT_DIR overrides; /tmp when /dev/shm is missing),
// mmap-ed by every tenant: a header and 62 slots of (pid, start time of that pid from /proc/<pid>/stat).  Registration,
// unregistration and pruning of dead entries run under flock(LOCK_EX); counting reads the mapping without a lock (a slot is one
// aligned 16-byte record whose pid word is written last / cleared first).  A process that dies without unregistering leaves its
// slot behind; whoever prunes next (every pyflex.set_scene, and every 64th pyflex.step) finds the pid gone -- or alive with another
// start time, i.e. reused -- and clears it.  Every pyflex.step in between reads the number of OCCUPIED slots (62 loads, no
// system call) and looks closer only when that number has changed: a worker learns of a neighbour one step after it registered.
// Processes in other containers (own pid namespace, own /dev/shm) or of other users are not seen: they have to say
// FLINGSIM_SHARED_GPU=1 themselves.
//
// Host-only code, no HIP: the CPU suite exercises it with real child processes (tests/test_pyflex_module.py).
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include <fcntl.h>
#include <signal.h>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/flingsim.h"
#include "fs_context.h"

namespace {

constexpr unsigned kMagic = 0x46535431u;  // "FST1"
constexpr int kSlots = 62;

struct Slot {
    volatile int pid;  // 0 = free
    int reserved;
    unsigned long long start;  // field 22 of /proc/<pid>/stat (clock ticks since boot): tells a reused pid from the registrant
};
struct Table {
    unsigned magic;
    unsigned version;
    unsigned reserved[2];
    Slot slot[kSlots];
};
static_assert(sizeof(Slot) == 16 && sizeof(Table) == 16 + 16 * kSlots, "table layout is shared between processes");

struct Mapping {
    std::string key;
    int fd = -1;
    Table *tab = nullptr;
    bool registered = false;  // this process asked to be listed (fs_tenants_register) and has not left since
};
Mapping g_map[4];  // a process rarely drives more than one device through the module; four is plenty
std::mutex g_map_mutex;  // (ctypes callers may come from several threads; the table itself is guarded by flock)

unsigned long long proc_start_time(int pid) {
    char path[64], buf[1024];
    snprintf(path, sizeof(path), "/proc/%d/stat", pid);
    FILE *fh = fopen(path, "r");
    if (!fh) return 0;
    size_t got = fread(buf, 1, sizeof(buf) - 1, fh);
    fclose(fh);
    buf[got] = 0;
    const char *p = strrchr(buf, ')');  // the command name may hold spaces and parentheses: fields resume after the LAST ')'
    if (!p) return 0;
    int field = 2;
    for (++p; *p; ++p) {
        if (*p == ' ') {
            ++field;
            if (field == 22) return strtoull(p + 1, nullptr, 10);
        }
    }
    return 0;
}

bool alive(const Slot &s, bool check_start) {
    const int pid = s.pid;
    if (pid <= 0) return false;
    if (kill(pid, 0) != 0 && errno == ESRCH) return false;
    if (check_start) {
        const unsigned long long st = proc_start_time(pid);
        if (st != 0 && s.start != 0 && st != s.start) return false;  // the pid belongs to somebody else now
    }
    return true;
}

std::string table_path(const char *device_key) {
    const char *dir = getenv("FLINGSIM_TENANT_DIR");
    struct stat st;
    if (!dir || !*dir) dir = (stat("/dev/shm", &st) == 0 && S_ISDIR(st.st_mode)) ? "/dev/shm" : "/tmp";
    std::string key;
    for (const char *p = device_key; *p; ++p) key += (isalnum((unsigned char)*p) || *p == '.' || *p == '-') ? *p : '_';
    return std::string(dir) + "/flingsim-tenants-" + std::to_string((unsigned)getuid()) + "-" + key;
}

Mapping *open_table(const char *device_key) {
    if (!device_key || !*device_key) {
        fs_set_error("tenant table: empty device key");
        return nullptr;
    }
    std::lock_guard<std::mutex> guard(g_map_mutex);
    Mapping *free_entry = nullptr;
    for (Mapping &m : g_map) {
        if (m.tab && m.key == device_key) return &m;
        if (!m.tab && !free_entry) free_entry = &m;
    }
    if (!free_entry) {
        fs_set_error("tenant table: more than 4 devices in one process");
        return nullptr;
    }
    const std::string path = table_path(device_key);
    int fd = open(path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0600);
    if (fd < 0) {
        fs_set_error("tenant table: cannot open " + path + ": " + strerror(errno));
        return nullptr;
    }
    if (flock(fd, LOCK_EX) != 0) {
        fs_set_error("tenant table: flock failed");
        close(fd);
        return nullptr;
    }
    struct stat st;
    bool fresh = fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(Table);
    if (fresh && ftruncate(fd, sizeof(Table)) != 0) {
        fs_set_error("tenant table: cannot size " + path);
        flock(fd, LOCK_UN);
        close(fd);
        return nullptr;
    }
    void *p = mmap(nullptr, sizeof(Table), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (p == MAP_FAILED) {
        fs_set_error("tenant table: mmap failed");
        flock(fd, LOCK_UN);
        close(fd);
        return nullptr;
    }
    Table *t = (Table *)p;
    if (fresh || t->magic != kMagic) {  // (a file of another layout: start over -- its registrants would not read ours either)
        memset(t, 0, sizeof(Table));
        t->version = 1;
        t->magic = kMagic;
    }
    flock(fd, LOCK_UN);
    free_entry->key = device_key;
    free_entry->fd = fd;
    free_entry->tab = t;
    return free_entry;
}

int count_live(Mapping *m, bool prune) {
    int n = 0;
    const bool locked = prune && flock(m->fd, LOCK_EX | LOCK_NB) == 0;  // somebody else is pruning: just count
    for (Slot &s : m->tab->slot) {
        if (s.pid <= 0) continue;
        if (alive(s, locked)) {
            ++n;
        } else if (locked) {
            s.pid = 0;
            s.start = 0;
        }
    }
    if (locked) flock(m->fd, LOCK_UN);
    return n;
}

}  // namespace

extern "C" int fs_tenants_register(const char *device_key) {
    Mapping *m = open_table(device_key);
    if (!m) return FS_ERR_STATE;
    const int me = (int)getpid();
    if (flock(m->fd, LOCK_EX) != 0) {
        fs_set_error("tenant table: flock failed");
        return FS_ERR_STATE;
    }
    Slot *mine = nullptr, *empty = nullptr;
    int n = 0;
    for (Slot &s : m->tab->slot) {
        if (s.pid == me) {
            mine = &s;  // registered before (pyflex.init after pyflex.clean, or a fork's parent entry: same pid only in the parent)
        } else if (s.pid > 0 && !alive(s, true)) {
            s.pid = 0;
            s.start = 0;
        }
        if (s.pid > 0) ++n;
        if (s.pid <= 0 && !empty) empty = &s;
    }
    if (!mine) {
        if (!empty) {  // 62 live tenants of one device: the answer to "is the device shared" is yes without us in the table
            flock(m->fd, LOCK_UN);
            return n + 1;
        }
        empty->start = proc_start_time(me);
        __sync_synchronize();
        empty->pid = me;
        ++n;
    } else {
        mine->start = proc_start_time(me);
    }
    m->registered = true;
    flock(m->fd, LOCK_UN);
    return n;
}

extern "C" int fs_tenants_unregister(const char *device_key) {
    Mapping *m = open_table(device_key);
    if (!m) return FS_ERR_STATE;
    const int me = (int)getpid();
    flock(m->fd, LOCK_EX);
    m->registered = false;
    for (Slot &s : m->tab->slot)
        if (s.pid == me) {
            s.pid = 0;
            s.start = 0;
        }
    flock(m->fd, LOCK_UN);
    return FS_OK;
}

extern "C" int fs_tenants_count(const char *device_key, int prune) {
    Mapping *m = open_table(device_key);
    if (!m) return FS_ERR_STATE;
    if (prune < 0) {  // occupied slots as they stand -- no system call at all: cheap enough for every pyflex.step
        int n = 0;
        for (const Slot &s : m->tab->slot) n += s.pid > 0 ? 1 : 0;
        return n;
    }
    if (prune > 0 && m->registered) {
        // self-heal: a tenant whose own entry has gone (a neighbour that could not see this pid -- another pid namespace on a
        // shared /dev/shm -- pruned it, or somebody reset the file) lists itself again before it counts
        const int me = (int)getpid();
        bool listed = false;
        for (const Slot &s : m->tab->slot) listed = listed || s.pid == me;
        if (!listed) (void)fs_tenants_register(device_key);
    }
    return count_live(m, prune != 0);
}
