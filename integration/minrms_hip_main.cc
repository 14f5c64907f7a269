// minrms_hip -- bin/minrms with the HIP engine behind the SearchOr interface.
// Same command line, same files written.  The flow below is the one of
// bin/minrms/main.cc:59-176 with the single change a maintainer makes: the
// search object is a SearchOrHipMinrms instead of a SearchOrDyn3d.
// MINRMS_BACKEND=cpu in the environment selects the reference's own
// SearchOrDyn3d again (same binary, for comparisons); anything else, or
// nothing, the HIP engine.
#include <vector>
#include <cassert>
#include <iostream>
using namespace std;

#include <stdlib.h>
#include <global_utils.h>
#include <parse_utils.h>
#include <or_gen.h>
#include "search_or.h"
#include "search_or_dyn3d.h"
#include "minrms_parser.h"
#include "search_or_hip_minrms.h"

const char g_version_string[] = "1.0.1";
const char g_date_string[]    = "<2019/10/13>";

using namespace minrms;
extern "C" void srand48(long);

int main(int argc, char **argv)
{
  const char *backend = getenv("MINRMS_BACKEND");
  const bool use_cpu = backend && string(backend) == "cpu";
  cout << "minrms v" << g_version_string << " " << g_date_string
       << (use_cpu ? " (CPU search)\n" : " (HIP engine)\n") << flush;
  srand48(42);
  vector<string> vOrigArgs, vArgs;
  ParseUtils::ConvertArgvToVectorOfStrings(argc, argv, vOrigArgs);
  char back_slash[2] = {'\\', '\0'};
  const char *env = getenv("MINRMS_SETTINGS_FILE");
  ParseUtils::HandleArgFiles(vOrigArgs, vArgs, "-cfg", "#", back_slash, env ? string(env) : string());
  MinrmsParser settings(vArgs);
  if (vArgs.size() != 1)
    ERR("Error: Unrecognized command line option: \"" << vArgs[1] << "\"\n");

  OrientationGenerator og(settings);
  SearchOr *pSearchOr = use_cpu ? static_cast<SearchOr *>(new SearchOrDyn3d(settings))
                                : static_cast<SearchOr *>(new SearchOrHipMinrms(settings));
  pSearchOr->Search(og, 0, og.size());
  if (settings.refine_method != SearchOr::Settings::NO_REFINE)
    pSearchOr->Refine();
  pSearchOr->WriteMSF();
  pSearchOr->WriteChimeraInfo();
  delete pSearchOr;
  return 0;
}
