% GpuTestOCProblem.m
% Shim for libocs.so (include/ocs.h); subclasses / replaces the reference's tests/TestOCProblem.m (OCProblem/OCProblem.m:8-21).
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_*.py).  See INTEGRATION.md.
classdef GpuTestOCProblem < OCProblem
   properties
      ControlBounds
      h            % libpointer to the ocs_problem handle
   end
   methods
      function obj = GpuTestOCProblem(p, ControlBounds)          % TestOCProblem.m:16
         obj.ControlBounds = ControlBounds;
         obj.h = libpointer('voidPtrPtr');
         ocs_check(calllib('libocs', 'ocs_problem_create', obj.h, 1, 1, 1, ...
                           [p.c p.m p.r], 3, ControlBounds(:)));
      end
      function value = F(obj, t, y, u)                           % OCProblem.m:12
         k = numel(t); value = zeros(size(y));
         [~, ~, ~, ~, ~, value] = calllib('libocs', 'ocs_problem_F', obj.h.Value, k, t, y, u, value);
      end
      function value = dFdx_times_vec(obj, t, y, u, v)           % OCProblem.m:16
         k = numel(t); value = zeros(size(y));
         [~, ~, ~, ~, ~, ~, value] = calllib('libocs', 'ocs_problem_dFdx_times_vec', ...
                                             obj.h.Value, k, t, y, u, v, value);
      end
      function value = dFdu_times_vec(obj, t, y, u, v)           % OCProblem.m:19
         k = numel(t); value = zeros(size(u));
         [~, ~, ~, ~, ~, ~, value] = calllib('libocs', 'ocs_problem_dFdu_times_vec', ...
                                             obj.h.Value, k, t, y, u, v, value);
      end
      function value = ControlChar(obj, t, x, lam)               % make_from_symbolic.m:33-38 (clamped, :111)
         k = numel(t); value = zeros(size(obj.ControlBounds, 1), k);
         [~, ~, ~, ~, value] = calllib('libocs', 'ocs_problem_ControlChar', obj.h.Value, k, t, x, lam, value);
      end
      function delete(obj), calllib('libocs', 'ocs_problem_destroy', obj.h.Value); end
   end
end
