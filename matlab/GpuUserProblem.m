% GpuUserProblem.m
% Shim for libocs.so (include/ocs.h): an OCProblem subclass (OCProblem/OCProblem.m:8-21) whose methods are given as
% device C++ source and compiled at run time with hipRTC (ocs_problem_create_from_source; contract of the source:
% optimal-control-solvers_amd/csrc/ocs_user_functor.hpp, examples: tests/user_problems.py).
% NOT VERIFIED: no MATLAB or Octave exists in the build pipeline; the same call sequence is tested through
% Python ctypes (tests/test_gpu_user_problems.py).  See INTEGRATION.md section 2b.
%
%   prob = GpuUserProblem(src, nS, nC, params, ControlBounds)            the three full-vector methods
%   prob = GpuUserProblem(..., 'ControlChar', true)                      the source defines ocs_ControlChar (fb_sweep)
%   prob = GpuUserProblem(..., 'RowFunctions', true)                     the source defines ocs_row_* (row-separable problem)
%   prob = GpuUserProblem(..., 'ControlFromCostate', true)               with both: ocs_ControlChar does not read x and
%                                                                        ocs_row_dFdy does not read u (two-kernel sweep)
classdef GpuUserProblem < OCProblem
   properties
      ControlBounds
      h            % libpointer to the ocs_problem handle
   end
   methods
      function obj = GpuUserProblem(src, nS, nC, params, ControlBounds, varargin)
         ip = inputParser;
         ip.addParameter('ControlChar', false);
         ip.addParameter('RowFunctions', false);
         ip.addParameter('ControlFromCostate', false);
         ip.parse(varargin{:});
         flags = 1*ip.Results.ControlChar + 2*ip.Results.RowFunctions + 4*ip.Results.ControlFromCostate;
         obj.ControlBounds = ControlBounds;
         obj.h = libpointer('voidPtrPtr');
         ocs_check(calllib('libocs', 'ocs_problem_create_from_source', obj.h, src, nS, nC, ...
                           params(:), numel(params), ControlBounds(:), flags));
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
