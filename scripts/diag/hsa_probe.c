/* Starts the HSA runtime the way every GPU process does (hsa_init) and, should that fault, prints a host back-trace.
 * Diagnostic for DESIGN.md 6a' (child GPU processes dying in hsa_init while a long-running parent lives).
 * gcc -O1 -g -rdynamic -o hsa_probe hsa_probe.c -ldl */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static void on_fault(int sig, siginfo_t* si, void* uc) {
    void* bt[64];
    char line[128];
    int n = snprintf(line, sizeof line, "hsa_probe: signal %d at address %p\n", sig, si ? si->si_addr : 0);
    (void)!write(2, line, n);
    n = backtrace(bt, 64);
    backtrace_symbols_fd(bt, n, 2);
    /* module map of the libraries in the trace, so offsets can be resolved afterwards */
    FILE* f = fopen("/proc/self/maps", "r");
    if (f) {
        char b[512];
        while (fgets(b, sizeof b, f))
            if (strstr(b, "r-xp") && (strstr(b, "hsa") || strstr(b, "libc") || strstr(b, "drm"))) fputs(b, stderr);
        fclose(f);
    }
    _exit(128 + sig);
}

int main(void) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = on_fault;
    sa.sa_flags = SA_SIGINFO;
    sigaction(SIGSEGV, &sa, 0);
    sigaction(SIGBUS, &sa, 0);
    sigaction(SIGABRT, &sa, 0);
    void* h = dlopen("libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/libhsa-runtime64.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { fprintf(stderr, "hsa_probe: dlopen failed: %s\n", dlerror()); return 3; }
    int (*init)(void) = (int (*)(void))dlsym(h, "hsa_init");
    int (*down)(void) = (int (*)(void))dlsym(h, "hsa_shut_down");
    if (!init) { fprintf(stderr, "hsa_probe: no hsa_init\n"); return 3; }
    int rc = init();
    printf("hsa_probe: hsa_init -> 0x%x\n", rc);
    if (rc == 0 && down) down();
    return rc ? 4 : 0;
}
