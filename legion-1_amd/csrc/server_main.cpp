// server_main.cpp -- the `legion` sampling-server binary (reference: src/main.cpp:4-9, started by
// legion_server.py:69 as `./src/legion <gpu_number> <cache_agg_mode>`; reads ./meta_config).
// Optional extras: argv[3] = comma-separated fan-outs (reference hard-codes 25,10,
// Server.cu:68-69), argv[4] = path of the meta_config file.
#include "../../include/legion_amd.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

int main(int argc, char** argv)
{
    if (argc < 3) {
        fprintf(stderr, "usage: %s <gpu_number> <cache_agg_mode> [fanouts e.g. 25,10,5] [meta_config path]\n", argv[0]);
        return 2;
    }
    const int gpus = atoi(argv[1]);
    const int mode = atoi(argv[2]);
    Server* server = NewGPUServer();
    if (argc >= 4 && argv[3][0]) {
        std::vector<int32_t> fan;
        char* dup = strdup(argv[3]);
        for (char* tok = strtok(dup, ","); tok; tok = strtok(nullptr, ",")) fan.push_back(atoi(tok));
        free(dup);
        if (!fan.empty()) Server_SetFanout(server, fan.data(), (int32_t)fan.size());
    }
    if (argc >= 5) Server_SetMetaConfigPath(server, argv[4]);
    Server_Initialize(server, gpus);
    if (legion_last_error()[0]) { fprintf(stderr, "%s\n", legion_last_error()); return 1; }
    Server_PreSc(server, mode);
    Server_Run(server);
    Server_Finalize(server);
    Server_Delete(server);
    int64_t audit[4] = {0, 0, 0, 0};
    legion_audit_counts(audit);
    return audit[1] > 0 ? 4 : 0;     // $LEGION_DEVICE_AUDIT=1: a wrong-device violation (listed by Server_Finalize) fails the run
}
