import ctypes, os, sys, subprocess, signal, tempfile, struct
lib_path = sys.argv[1]
d = tempfile.mkdtemp(); os.environ["FLINGSIM_TENANT_DIR"] = d
lib = ctypes.CDLL(lib_path)
for f in ("fs_tenants_register", "fs_tenants_unregister"): getattr(lib, f).argtypes = [ctypes.c_char_p]
lib.fs_tenants_count.argtypes = [ctypes.c_char_p, ctypes.c_int]
key = b"0000:05:00.0"
assert lib.fs_tenants_register(key) == 1
child = subprocess.Popen([sys.executable, "-c", "import ctypes,sys,time; l=ctypes.CDLL(sys.argv[1]); l.fs_tenants_register.argtypes=[ctypes.c_char_p]; print(l.fs_tenants_register(sys.argv[2].encode()), flush=True); time.sleep(30)", lib_path, key.decode()], stdout=subprocess.PIPE, text=True)
assert int(child.stdout.readline()) == 2
child.send_signal(signal.SIGKILL); child.wait()
assert lib.fs_tenants_count(key, -1) == 2 and lib.fs_tenants_count(key, 1) == 1
for k in (b"a", b"b", b"c"): assert lib.fs_tenants_count(k, 1) == 0
assert lib.fs_tenants_count(b"one too many", 1) < 0          # more than 4 devices in one process: refused, not overrun
assert lib.fs_tenants_unregister(key) == 0 and lib.fs_tenants_count(key, 1) == 0
print("tenant table under ASan/UBSan: ok")
