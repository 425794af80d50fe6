% ocs_load.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's library loading step.
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
function ocs_load(root)
% ocs_load(root): load libocs once; root = checkout directory of this repository
   if nargin < 1, root = fileparts(fileparts(mfilename('fullpath'))); end
   if ~libisloaded('libocs')
      loadlibrary(fullfile(root, 'optimal-control-solvers_amd', 'lib', 'libocs.so'), ...
                  fullfile(root, 'include', 'ocs.h'), 'alias', 'libocs');
   end
end
