% ocs_check.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's error convention (negative status -> MATLAB error).
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
function ocs_check(rc)
   if rc < 0, error('libocs:%d %s', rc, calllib('libocs', 'ocs_last_error')); end
end
