// audit_hooks.h -- routes the HIP resource calls of a source file through the logical-device audit (audit.h).
// Include it LAST (behind every system / library header: the names below are function-like macros from here on).  With the audit
// off each wrapper is the HIP call behind one branch.
#pragma once
#include "audit.h"

#define hipMalloc(pp, n)                        ::legion::audit::Malloc((void**)(pp), (n), __FILE__, __LINE__)
#define hipMallocManaged(pp, n)                 ::legion::audit::MallocManaged((void**)(pp), (n), __FILE__, __LINE__)
#define hipFree(p)                              ::legion::audit::Free((void*)(p), __FILE__, __LINE__)
#define hipHostMalloc(pp, n, flags)             ::legion::audit::HostMalloc((void**)(pp), (n), (flags), __FILE__, __LINE__)
#define hipHostFree(p)                          ::legion::audit::HostFree((void*)(p), __FILE__, __LINE__)
#define hipIpcOpenMemHandle(pp, h, flags)       ::legion::audit::IpcOpen((void**)(pp), (h), (flags), __FILE__, __LINE__)
#define hipIpcCloseMemHandle(p)                 ::legion::audit::IpcClose((void*)(p), __FILE__, __LINE__)
#define hipStreamCreateWithFlags(ps, flags)     ::legion::audit::StreamCreateWithFlags((ps), (flags), __FILE__, __LINE__)
#define hipStreamCreateWithPriority(ps, fl, pr) ::legion::audit::StreamCreateWithPriority((ps), (fl), (pr), __FILE__, __LINE__)
#define hipExtStreamCreateWithCUMask(ps, n, m)  ::legion::audit::StreamCreateWithCUMask((ps), (n), (m), __FILE__, __LINE__)
#define hipStreamDestroy(s)                     ::legion::audit::StreamDestroy((s), __FILE__, __LINE__)
#define hipEventCreate(pe)                      ::legion::audit::EventCreate((pe), __FILE__, __LINE__)
#define hipEventCreateWithFlags(pe, flags)      ::legion::audit::EventCreateWithFlags((pe), (flags), __FILE__, __LINE__)
#define hipEventDestroy(e)                      ::legion::audit::EventDestroy((e), __FILE__, __LINE__)
#define hipEventRecord(e, s)                    ::legion::audit::EventRecord((e), (s), __FILE__, __LINE__)
#define hipStreamWaitEvent(s, e, flags)         ::legion::audit::StreamWaitEvent((s), (e), (flags), __FILE__, __LINE__)
#define hipMemcpy(d, s, n, kind)                ::legion::audit::Memcpy((d), (s), (n), (kind), __FILE__, __LINE__)
#define hipMemcpyAsync(d, s, n, kind, st)       ::legion::audit::MemcpyAsync((d), (s), (n), (kind), (st), __FILE__, __LINE__)
#define hipMemcpy2D(d, dp, s, sp, w, h, kind)   ::legion::audit::Memcpy2D((d), (dp), (s), (sp), (w), (h), (kind), __FILE__, __LINE__)
#define hipMemcpyPeerAsync(d, dd, s, sd, n, st) ::legion::audit::MemcpyPeerAsync((d), (dd), (s), (sd), (n), (st), __FILE__, __LINE__)
#define hipMemset(d, v, n)                      ::legion::audit::Memset((d), (v), (n), __FILE__, __LINE__)
#define hipMemsetAsync(d, v, n, st)             ::legion::audit::MemsetAsync((d), (v), (n), (st), __FILE__, __LINE__)
#define hipGraphInstantiate(pe, g, en, log, n)  ::legion::audit::GraphInstantiate((pe), (g), (en), (log), (n), __FILE__, __LINE__)
#define hipGraphLaunch(e, s)                    ::legion::audit::GraphLaunch((e), (s), __FILE__, __LINE__)
#define hipGraphExecDestroy(e)                  ::legion::audit::GraphExecDestroy((e), __FILE__, __LINE__)
#define hipStreamBeginCapture(s, mode)          ::legion::audit::StreamBeginCapture((s), (mode), __FILE__, __LINE__)
