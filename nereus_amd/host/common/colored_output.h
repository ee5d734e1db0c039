// Terminal colours for the solvers' one-line status messages.  Only GREEN and RESET are used by the solver
// classes (as in the reference, sph.cpp:32 / iisph.cpp:31); the escape sequences are composed from the SGR code so
// callers that include this header for other colours (the reference header offers the 8 ANSI colours and their
// bold forms under the same names) keep compiling.
#pragma once
#define NRS_SGR(code) "\033[" code "m"
#define NRS_SGR_BOLD(code) NRS_SGR("1") NRS_SGR(code)
#define RESET NRS_SGR("0")
#define GREEN NRS_SGR("32")
#define RED NRS_SGR("31")
#define YELLOW NRS_SGR("33")
#define BLUE NRS_SGR("34")
#define MAGENTA NRS_SGR("35")
#define CYAN NRS_SGR("36")
#define WHITE NRS_SGR("37")
#define BLACK NRS_SGR("30")
#define BOLDGREEN NRS_SGR_BOLD("32")
#define BOLDRED NRS_SGR_BOLD("31")
#define BOLDYELLOW NRS_SGR_BOLD("33")
#define BOLDBLUE NRS_SGR_BOLD("34")
#define BOLDMAGENTA NRS_SGR_BOLD("35")
#define BOLDCYAN NRS_SGR_BOLD("36")
#define BOLDWHITE NRS_SGR_BOLD("37")
#define BOLDBLACK NRS_SGR_BOLD("30")
